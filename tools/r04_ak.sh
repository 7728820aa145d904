#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
python -c "from hotformerloc_amd import _native; _native.load(); print('library ok')" || exit 1
timeout 120 python tools/hop_latency.py 2>&1 | tail -4 | tee gpurun_out/r04_ak_hop.log
