#!/bin/bash
# last check of the final tree: smoke(), golden model tests, default bench line without the CPU baseline
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r06_ab_smoke.log 2>&1; tail -3 gpurun_out/r06_ab_smoke.log
timeout 1500 python -m pytest tests/test_gpu_model.py -q -m gpu -k "golden or schedule or executor" > gpurun_out/r06_ab_model_tests.log 2>&1; tail -3 gpurun_out/r06_ab_model_tests.log
timeout 300 python bench.py --no-extras --no-cpu-baseline > gpurun_out/r06_ab_bench.json 2>/dev/null; cut -c1-300 gpurun_out/r06_ab_bench.json
