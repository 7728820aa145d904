set -x
export HFL_PROBES=1   # the HFL_* schedule knobs below are probe switches (hotformerloc_amd/model.py)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
for ws in 1 0; do
export HFL_ATTN_WS=$ws
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r05_n_trace$ws -- python bench.py --steps 6 --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/r05_n_trace$ws.log 2>&1
tr=$(find gpurun_out/r05_n_trace$ws -name '*kernel_trace.csv' | head -1)
python tools/forward_phases.py "$tr" "iteration 5" > gpurun_out/r05_n_phases_ws$ws.log 2>&1
rm -rf gpurun_out/r05_n_trace$ws
done
