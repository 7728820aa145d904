"""One H-OSA iteration of the last traced forward, all queues, from a rocprofv3 --kernel-trace CSV:
python tools/iteration_timeline.py <kernel_trace.csv> [iteration index]"""
import csv
import re
import sys


def short(name):
    m = re.search(r'\(anonymous namespace\)::(\w+)', name)
    if m:
        t = re.search(r'<([^>]*)>', name)
        return m.group(1) + ('<%s>' % t.group(1) if t else '')
    for k, v in (('Cijk_', 'hipBLASLt'), ('CatArray', 'cat'), ('copyBuffer', 'copy'), ('elementwise', 'eltwise'),
                 ('fillBuffer', 'fill')):
        if k in name:
            return v
    return name[:30]


rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
it = int(sys.argv[2]) if len(sys.argv) > 2 else 5
# forwards end with the head's three segment_softmax launches; take the last complete forward
heads = [i for i, r in enumerate(rows) if 'segment_softmax' in r['Kernel_Name']]
end = heads[-1]
start = heads[-4] + 1
fwd = rows[start:end + 1]
t0 = int(fwd[0]['Start_Timestamp'])
t1 = max(int(r['End_Timestamp']) for r in fwd)
print('last forward: %d kernels, %.3f ms' % (len(fwd), (t1 - t0) / 1e6))
ra = [r for r in fwd if 'relay_attn' in r['Kernel_Name']]
a = int(ra[it]['Start_Timestamp']) - 150000
b = int(ra[it + 1]['Start_Timestamp']) - 150000
print('iteration %d: %.1f us' % (it, (b - a) / 1e3))
last_end = {}
for r in fwd:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if a <= s < b:
        q = r['Queue_Id']
        gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
        print(' q%s %8.1f us +%7.1f us (gap %6.1f) %-38s grid %s' % (q, (s - a) / 1e3, (e - s) / 1e3, gap, short(r['Kernel_Name']),
                                                                      r['Grid_Size_X']))
    last_end[r['Queue_Id']] = e
