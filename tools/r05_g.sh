set -x
export HFL_PROBES=1   # the HFL_* schedule knobs below are probe switches (hotformerloc_amd/model.py)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python tools/host_issue_probe.py > gpurun_out/r05_g_graph_probe.log 2>&1
# kernel trace of the current default schedule -> iteration timeline
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r05_g_trace -- python bench.py --steps 6 --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/r05_g_trace.log 2>&1
tr=$(find gpurun_out/r05_g_trace -name '*kernel_trace.csv' | head -1)
python tools/forward_phases.py "$tr" "iteration 5" > gpurun_out/r05_g_phases.log 2>&1
rm -rf gpurun_out/r05_g_trace
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 40 --warmup 10 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$tag', d['value'], d['ms_per_step'])" >> gpurun_out/r05_g_ab.log; }
for i in 1 2 3; do
run A X=1
run B HFL_MLP_FUSED_MIN_ROWS=1000 HFL_QKV_FUSED_MIN_ROWS=1000
run C HFL_RTSA_STREAM=0
run D HFL_MERGED_ATTN=0
run F HFL_MLP_FUSED_MIN_ROWS=1000 HFL_QKV_FUSED_MIN_ROWS=1000 HFL_RTSA_STREAM=0
done
