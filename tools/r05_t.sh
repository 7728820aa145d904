set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python tools/train_ops_profile.py > gpurun_out/r05_t_train_ops.txt 2> gpurun_out/r05_t_train_ops.err
