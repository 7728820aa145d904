set -x
export HFL_PROBES=1   # the HFL_* schedule knobs below are probe switches (hotformerloc_amd/model.py)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/r05_y_ab.log
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 40 --warmup 10 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$tag', d['value'], d['ms_per_step'])" >> gpurun_out/r05_y_ab.log; }
for i in 1 2 3; do
run default X=1
run ws_d3 HFL_ATTN_WS_MIN_ROWS=10000
run coarse_unfused HFL_MLP_FUSED_MIN_ROWS=24576 HFL_QKV_FUSED_MIN_ROWS=24576
run coarse_qkv_unfused HFL_QKV_FUSED_MIN_ROWS=24576
run no_side_streams HFL_PYRAMID_STREAMS=0
done
