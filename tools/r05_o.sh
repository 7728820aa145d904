set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "attn_ws" -s 2>&1 | grep -v amdgpu.ids | tail -20 > gpurun_out/r05_o_test.log
