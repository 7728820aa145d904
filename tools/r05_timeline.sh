set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r05_tl_trace -- python bench.py --steps 6 --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/r05_tl_trace.log 2>&1
tr=$(find gpurun_out/r05_tl_trace -name '*kernel_trace.csv' | head -1)
python tools/forward_phases.py "$tr" "iteration 5" > gpurun_out/r05_phases_final.log 2>&1
rm -rf gpurun_out/r05_tl_trace
