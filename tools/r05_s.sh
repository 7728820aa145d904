set -x
export HFL_PROBES=1   # the HFL_* schedule knobs below are probe switches (hotformerloc_amd/model.py)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_loss.py tests/test_gpu_configs.py -x -q -m gpu -k "checkpoint or forward_backward or multistaged or rccl or drop" 2>&1 | grep -v amdgpu.ids | tail -8 > gpurun_out/r05_s_test.log
timeout 900 python bench.py --config cs-wild-places --train --multistaged --steps 5 --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/r05_s_multistaged.json 2> gpurun_out/r05_s_multistaged.err
HFL_CHECKPOINT=always timeout 900 python bench.py --config cs-wild-places --train --multistaged --steps 5 --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/r05_s_multistaged_always.json 2>> gpurun_out/r05_s_multistaged.err
