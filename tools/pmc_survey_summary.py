"""Mean per-launch value of every counter collected by tools/pmc_survey.sh for the window kernel."""
import csv, glob, collections, sys
tag = sys.argv[1]
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob('gpurun_out/survey_%s_g*/**/*counter_collection.csv' % tag, recursive=True):
    for r in csv.DictReader(open(f)):
        if 'window_attn' not in r['Kernel_Name']:
            continue
        a = agg[r['Counter_Name']]; a[0] += 1; a[1] += float(r['Counter_Value'])
for k in sorted(agg):
    n, v = agg[k]
    print('%-32s %14.0f  (%d launches)' % (k, v / n, n))
