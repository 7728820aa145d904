set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python bench.py --config cs-wild-places --train --steps 5 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | head -c 400 > gpurun_out/r05_e_train.json
timeout 1500 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "forward_backward or checkpoint" 2>&1 | tail -8 > gpurun_out/r05_e_test.log
