set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
HFL_RT_ABLATE=1 timeout 300 python tools/attn_fused_rt_probe.py > gpurun_out/r05_b_probe.log 2>&1
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "attn_fused" 2>&1 | tail -25 > gpurun_out/r05_b_test.log
