#!/bin/bash
# SQ counter surveys of the two row-tile GEMM kernels at the depth-4 shape; kernel shapes of the one-stream schedule
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
out=gpurun_out
python -c "from hotformerloc_amd import _native; _native.load(); print('library ok')" || exit 1
tools/pmc_survey2.sh r04_mlp ln_mlp_fused_kernel tools/mlp_fused_one.py 65536 256 > $out/r04_mlp_counters.txt 2>&1
rm -rf $out/survey_r04_mlp_g*
rocprofv3 --kernel-trace --output-format csv -d $out/r04_p_trace -- python bench.py --serial-streams --steps 8 --warmup 2 --no-extras --no-cpu-baseline > $out/r04_p_trace.log 2>&1
f=$(find $out/r04_p_trace -name "*kernel_trace.csv" | head -1)
python tools/kernel_shapes.py $f 10 > $out/r04_p_serial_shapes.md
rm -rf $out/r04_p_trace
cat $out/r04_mlp_counters.txt
head -45 $out/r04_p_serial_shapes.md
