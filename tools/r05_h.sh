set -x
export HFL_PROBES=1   # the HFL_* schedule knobs below are probe switches (hotformerloc_amd/model.py)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "golden or native_block or early_phase or merged" 2>&1 | tail -6 > gpurun_out/r05_h_test.log
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 40 --warmup 10 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$tag', d['value'], d['ms_per_step'])" >> gpurun_out/r05_h_ab.log; }
for i in 1 2 3; do
run defer1 X=1
run defer0 HFL_DEFER_QKV=0
run defer1_rows24k HFL_MLP_FUSED_MIN_ROWS=24576 HFL_QKV_FUSED_MIN_ROWS=24576
done
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r05_h_trace -- python bench.py --steps 6 --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/r05_h_trace.log 2>&1
tr=$(find gpurun_out/r05_h_trace -name '*kernel_trace.csv' | head -1)
python tools/forward_phases.py "$tr" "iteration 5" > gpurun_out/r05_h_phases.log 2>&1
rm -rf gpurun_out/r05_h_trace
