set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r06_g_gputest.log 2>&1
tail -5 gpurun_out/r06_g_gputest.log
bash tools/r06_e.sh
