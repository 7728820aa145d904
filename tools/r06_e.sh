set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python bench.py --no-cpu-baseline --no-train-leg --no-oxford-leg --no-pinned-leg > gpurun_out/r06_e_bench.json 2> gpurun_out/r06_e_bench.err
tail -3 gpurun_out/r06_e_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r06_e_prof_x6 -- python bench.py --gemm x6 --no-extras --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/r06_e_prof_x6.log 2>&1
find gpurun_out/r06_e_prof_x6 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r06_e_x6_kernel_stats.csv
python tools/summarize_rocprof.py gpurun_out/r06_e_x6_kernel_stats.csv 13 > gpurun_out/r06_e_x6_summary_table.md 2>&1 || true
head -40 gpurun_out/r06_e_x6_summary_table.md
rm -rf gpurun_out/r06_e_prof_x6
