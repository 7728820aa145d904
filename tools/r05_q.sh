set -x
export HFL_PROBES=1   # the HFL_* schedule knobs below are probe switches (hotformerloc_amd/model.py)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/r05_q_ab.log
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 40 --warmup 10 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$tag', d['value'], d['ms_per_step'])" >> gpurun_out/r05_q_ab.log; }
for i in 1 2 3; do
run both X=1
run noseg HFL_RTSA_SEGMENTS=0
run nocopyless HFL_RELAY_IN_PLACE=0
run neither HFL_RELAY_IN_PLACE=0 HFL_RTSA_SEGMENTS=0
done
