set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "proj_mlp_fused or ln_mlp_fused" 2>&1 | tail -15 > gpurun_out/r05_a_test.log
timeout 300 python tools/proj_mlp_probe.py > gpurun_out/r05_a_probe.log 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r05_a_bench.json 2> gpurun_out/r05_a_bench.err
tail -c 600 gpurun_out/r05_a_bench.err
