#!/bin/bash
# round-4 evidence, part B: SQ / TCC counter surveys of the attention kernels, training profile
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
out=gpurun_out
python -c "from hotformerloc_amd import _native; _native.load(); print('library ok')" || exit 1
tools/pmc_survey2.sh r04_attn window_attn_kernel_v5 tools/attn_v5_one.py 4 4 > $out/r04_attn_counters.txt 2>&1
tools/pmc_survey2.sh r04_fused attn_fused_kernel tools/attn_fused_probe.py > $out/r04_fused_counters.txt 2>&1
tools/pmc_survey2.sh r04_mlp ln_mlp_fused_kernel tools/mlp_fused_one.py 65536 256 > $out/r04_mlp_counters.txt 2>&1
rm -rf $out/survey_r04_attn_g* $out/survey_r04_fused_g* $out/survey_r04_mlp_g*
tools/prof_train.sh r04_train_cs --config cs-wild-places > $out/r04_train_prof.log 2>&1
rm -rf $out/r04_train_cs_stats
cat $out/r04_attn_counters.txt | tail -60
cat $out/r04_fused_counters.txt | tail -60
head -30 $out/r04_train_cs_summary_table.md
