#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "relay" > gpurun_out/r06_w_kernel_tests.log 2>&1; tail -3 gpurun_out/r06_w_kernel_tests.log
timeout 900 python tools/train_ops_profile.py > gpurun_out/r06_w_train_ops.txt 2> gpurun_out/r06_w_train_ops.err; head -90 gpurun_out/r06_w_train_ops.txt
