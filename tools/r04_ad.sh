#!/bin/bash
# A/B: the bench process pinned to 1/8 of the host's logical CPUs (the share of one of 8 ranks) against unpinned
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
out=gpurun_out
python -c "from hotformerloc_amd import _native; _native.load(); print('library ok')" || exit 1
nproc; lscpu | grep -i "numa\|socket\|model name" | head -8
run() { # label, args...
  label=$1; shift
  python bench.py --no-extras --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-34s' % '$label', j['value'], j['ms_per_step'], j['host_issue']['ms_per_step_issue'])"
}
for i in 1 2 3 4; do
  run "unpinned"
  run "pinned to 32 CPUs" --pin-cores 32
  run "pinned to 8 CPUs" --pin-cores 8
done > $out/r04_ad_ab.log 2>&1
cat $out/r04_ad_ab.log
