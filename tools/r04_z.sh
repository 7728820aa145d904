#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
out=gpurun_out
python -c "from hotformerloc_amd import _native; _native.load(); print('library ok')" || exit 1
timeout 300 python tools/mlp_ablate.py 65536 > $out/r04_z_ablate.log 2>&1; cat $out/r04_z_ablate.log | tail -15
