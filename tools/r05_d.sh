set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "full_size" 2>&1 | tail -8 > gpurun_out/r05_d_test.log
timeout 900 python tools/train_ops_profile.py > gpurun_out/r05_d_train_ops.txt 2> gpurun_out/r05_d_train_ops.err
tail -3 gpurun_out/r05_d_train_ops.err
