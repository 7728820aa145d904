#!/bin/bash
# fused LN -> qkv: 4 waves x 2 row tiles against 8 waves x 1
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
out=gpurun_out
python -c "from hotformerloc_amd import _native; _native.load(); print('library ok')" || exit 1
timeout 600 python tools/mlp_waves_probe.py > $out/r04_u_probe.log 2>&1; cat $out/r04_u_probe.log | tail -12
