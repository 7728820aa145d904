set -x
export HFL_PROBES=1   # the HFL_* schedule knobs below are probe switches (hotformerloc_amd/model.py)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "relay_attention" 2>&1 | tail -15 > gpurun_out/r05_c_test.log
timeout 1200 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "golden or native_block or full_size" 2>&1 | tail -15 >> gpurun_out/r05_c_test.log
for i in 1 2; do
HFL_RTSA_SLIM=0 timeout 600 python bench.py --steps 40 --warmup 10 --no-extras --no-cpu-baseline 2>/dev/null | head -c 300 > gpurun_out/r05_c_bench_slim0_$i.json
HFL_RTSA_SLIM=1 timeout 600 python bench.py --steps 40 --warmup 10 --no-extras --no-cpu-baseline 2>/dev/null | head -c 300 > gpurun_out/r05_c_bench_slim1_$i.json
done
