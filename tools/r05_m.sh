set -x
export HFL_PROBES=1   # the HFL_* schedule knobs below are probe switches (hotformerloc_amd/model.py)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 40 --warmup 10 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$tag', d['value'], d['ms_per_step'])" >> gpurun_out/r05_m_ab.log; }
echo "# box $(hostname) $(date +%s)" >> gpurun_out/r05_m_ab.log
for i in 1 2 3; do
run ws0 HFL_ATTN_WS=0
run ws1 HFL_ATTN_WS=1
run ws1_noearly HFL_ATTN_WS=1 HFL_EARLY_PHASE=0
run ws0_noearly HFL_ATTN_WS=0 HFL_EARLY_PHASE=0
done
