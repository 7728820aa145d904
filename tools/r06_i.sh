set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python tools/x6p_probe.py > gpurun_out/r06_i_x6p_probe.log 2>&1
cat gpurun_out/r06_i_x6p_probe.log
