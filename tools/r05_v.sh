set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python tools/attn_ws_probe.py > gpurun_out/r05_v_probe.log 2>&1
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "attn_ws" 2>&1 | tail -3 > gpurun_out/r05_v_test.log
