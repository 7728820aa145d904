set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python bench.py --no-cpu-baseline --no-train-leg --no-oxford-leg --no-pinned-leg > gpurun_out/r05_z_bench.json 2> gpurun_out/r05_z_bench.err
