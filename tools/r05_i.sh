set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "early_phase or native_block" 2>&1 | tail -4 > gpurun_out/r05_i_test.log
timeout 1500 bash tools/cpe_counters.sh r05_i
