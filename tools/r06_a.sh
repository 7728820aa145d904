set -x
export HFL_PROBES=1   # the HFL_* schedule knobs below are probe switches (hotformerloc_amd/model.py)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python tools/x6_probe.py > gpurun_out/r06_a_x6_probe.log 2>&1
timeout 900 python -m pytest tests -m gpu -x -q -k "cpe or dwconv" > gpurun_out/r06_a_cpe_tests.log 2>&1
timeout 900 bash tools/cpe_counters.sh r06_a_xcd
export HFL_VARIANTS=cpe_chunk_rows=0
timeout 900 bash tools/cpe_counters.sh r06_a_interleaved
unset HFL_VARIANTS
for i in 1 2; do
  timeout 600 python bench.py --no-cpu-baseline --no-train-leg --no-oxford-leg --no-pinned-leg > gpurun_out/r06_a_bench_xcd_$i.json 2> gpurun_out/r06_a_bench_xcd_$i.err
  HFL_VARIANTS=cpe_chunk_rows=0 timeout 600 python bench.py --no-cpu-baseline --no-train-leg --no-oxford-leg --no-pinned-leg > gpurun_out/r06_a_bench_il_$i.json 2> gpurun_out/r06_a_bench_il_$i.err
done
tail -3 gpurun_out/r06_a_cpe_tests.log
cat gpurun_out/r06_a_x6_probe.log
