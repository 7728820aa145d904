#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
out=gpurun_out
python -c "from hotformerloc_amd import _native; _native.load(); print('library ok')" || exit 1
timeout 300 python tools/mlp_ablate.py 65536 > $out/r04_aa_ablate.log 2>&1; cat $out/r04_aa_ablate.log | tail -8
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "mlp_fused or attn_pool" > $out/r04_aa_tests.log 2>&1; tail -5 $out/r04_aa_tests.log
timeout 300 python tools/mlp_waves_probe.py lag > $out/r04_aa_probe.log 2>&1; cat $out/r04_aa_probe.log | tail -10 | cut -c1-200
