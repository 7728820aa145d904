#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/pmc_survey2.sh r06_attn_bwd "window_attn_bwd2_kernel" tools/attn_bwd_one.py > gpurun_out/r06_attn_bwd_counters.txt 2>&1
rm -rf gpurun_out/survey_r06_attn_bwd_g*
cat gpurun_out/r06_attn_bwd_counters.txt
