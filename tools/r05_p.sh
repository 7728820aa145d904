set -x
export HFL_PROBES=1   # the HFL_* schedule knobs below are probe switches (hotformerloc_amd/model.py)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/r05_p_ab.log
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 40 --warmup 10 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$tag', d['value'], d['ms_per_step'])" >> gpurun_out/r05_p_ab.log; }
for i in 1 2 3; do
run default X=1
run ws_early HFL_ATTN_WS=1
run ws_noearly HFL_ATTN_WS=1 HFL_EARLY_PHASE=0
run ws_noearly_d3 HFL_ATTN_WS=1 HFL_EARLY_PHASE=0 HFL_ATTN_WS_MIN_ROWS=10000
run ws_noearly_nomerge HFL_ATTN_WS=1 HFL_EARLY_PHASE=0 HFL_MERGED_ATTN=0
run ws_noearly_nostreams HFL_ATTN_WS=1 HFL_EARLY_PHASE=0 HFL_PYRAMID_STREAMS=0
done
