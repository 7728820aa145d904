set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "x6" > gpurun_out/r06_d_x6_tests.log 2>&1
tail -15 gpurun_out/r06_d_x6_tests.log
timeout 900 python tools/x6_probe.py > gpurun_out/r06_d_x6_probe.log 2>&1
cat gpurun_out/r06_d_x6_probe.log
