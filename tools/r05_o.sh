set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_kernels.py -x -q -m gpu -k "attn_ws or early_phase or native_block" 2>&1 | tail -8 > gpurun_out/r05_o_test.log
timeout 300 python tools/attn_ws_probe.py > gpurun_out/r05_o_probe.log 2>&1
