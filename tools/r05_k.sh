set -x
export HFL_PROBES=1   # the HFL_* schedule knobs below are probe switches (hotformerloc_amd/model.py)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
HFL_WS_ABLATE=1 timeout 300 python tools/attn_ws_probe.py > gpurun_out/r05_k_probe.log 2>&1
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "attn_ws" 2>&1 | tail -15 > gpurun_out/r05_k_test.log
