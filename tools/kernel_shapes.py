"""Kernel time by (kernel, grid size) from a rocprofv3 --kernel-trace CSV: which SHAPES of a kernel carry its time.
    python tools/kernel_shapes.py <kernel_trace.csv> <steps> [kernel name substring ...]"""
import collections
import csv
import re
import sys


def short(name):
    m = re.search(r'\(anonymous namespace\)::(\w+)', name)
    if m:
        t = re.search(r'<([^>]*)>', name)
        return m.group(1) + ('<%s>' % t.group(1) if t else '')
    for k, v in (('Cijk_', 'hipBLASLt'), ('CatArray', 'cat'), ('copyBuffer', 'copy'), ('elementwise', 'eltwise'),
                 ('fillBuffer', 'fill'), ('index', 'index')):
        if k in name:
            return v
    return name[:28]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    steps = float(sys.argv[2])
    want = sys.argv[3:]
    agg = collections.defaultdict(lambda: [0, 0])
    for r in rows:
        n = short(r['Kernel_Name'])
        if want and not any(w in n for w in want):
            continue
        key = (n, int(r.get('Grid_Size_X', r.get('Grid_Size', 0))), int(r.get('Workgroup_Size_X', r.get('Workgroup_Size', 0))))
        a = agg[key]
        a[0] += 1
        a[1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    tot = sum(v[1] for v in agg.values())
    print('| kernel | grid (threads) | workgroups | launches/step | avg us | ms/step | % of listed |')
    print('|---|---|---|---|---|---|---|')
    for (n, g, wg), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
        print('| %s | %d | %d | %.1f | %.1f | %.3f | %.1f |' % (n, g, g // max(wg, 1), c / steps, t / c / 1e3, t / steps / 1e6,
                                                                100.0 * t / max(tot, 1)))


if __name__ == '__main__':
    main()
