set -x
export HFL_PROBES=1   # the HFL_* schedule knobs below are probe switches (hotformerloc_amd/model.py)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/r05_r_ab.log
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --config cs-wild-places --train --steps 5 --warmup 3 --no-extras --no-cpu-baseline 2>gpurun_out/r05_r_err_$tag.log | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$tag', d['value'], d['ms_per_step'], d.get('peak_memory_GiB'))" >> gpurun_out/r05_r_ab.log; }
for i in 1 2; do
run always HFL_CHECKPOINT=always
run never HFL_CHECKPOINT=never
run auto HFL_CHECKPOINT=auto
done
