#!/bin/bash
# one rocprofv3 --pmc run per counter group given on the command line:
#   tools/pmc_groups.sh <tag> <kernel name substring> "<group 1 counters>" ["<group 2>" ...] -- <script> [args...]
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
tag=$1; kern=$2; shift; shift
groups=()
while [ "$1" != "--" ]; do groups+=("$1"); shift; done
shift
i=0
for g in "${groups[@]}"; do
  i=$((i+1))
  rocprofv3 --pmc $g --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}_g$i -- python "$@" > gpurun_out/pmc_${tag}_g$i.log 2>&1
done
python - "$tag" "$kern" <<'PY'
import csv, glob, collections, sys
tag, kern = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob('gpurun_out/pmc_%s_g*/**/*counter_collection.csv' % tag, recursive=True):
    for r in csv.DictReader(open(f)):
        if kern not in r['Kernel_Name']:
            continue
        a = agg[r['Counter_Name']]; a[0] += 1; a[1] += float(r['Counter_Value'])
for k in sorted(agg):
    n, v = agg[k]
    print('%-32s %16.0f  (%d samples)' % (k, v / n, n))
PY
