"""Critical-path view of one forward from a rocprofv3 --kernel-trace CSV (kernel_trace.csv with Start/End timestamps and
Stream/Queue ids): per queue busy time, idle gaps, and the kernels of the longest queue in order.
usage: python tools/timeline_probe.py <kernel_trace.csv> [first_step last_step]"""
import collections
import csv
import re
import sys


def short(name):
    m = re.search(r'\(anonymous namespace\)::(\w+)', name)
    if m:
        t = re.search(r'<([^>]*)>', name)
        return m.group(1) + ('<%s>' % t.group(1) if t else '')
    if name.startswith('Cijk_'):
        return 'hipBLASLt'
    return name[:40]


def main(path):
    rows = list(csv.DictReader(open(path)))
    key_q = 'Queue_Id' if 'Queue_Id' in rows[0] else 'Stream_Id'
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    t0, t1 = int(rows[0]['Start_Timestamp']), max(int(r['End_Timestamp']) for r in rows)
    print('kernels %d, span %.3f ms' % (len(rows), (t1 - t0) / 1e6))
    per_q = collections.defaultdict(list)
    for r in rows:
        per_q[r[key_q]].append((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name'])))
    for q, ks in sorted(per_q.items(), key=lambda kv: -sum(e - s for s, e, _ in kv[1])):
        busy = sum(e - s for s, e, _ in ks)
        gaps = [ks[i + 1][0] - ks[i][1] for i in range(len(ks) - 1)]
        big = sorted(gaps)[-5:] if gaps else []
        print('queue %s: %d kernels, busy %.3f ms, gaps>5us: %d (sum %.3f ms), largest gaps us %s'
              % (q, len(ks), busy / 1e6, sum(1 for g in gaps if g > 5000), sum(g for g in gaps if g > 5000) / 1e6,
                 [round(g / 1e3, 1) for g in big]))
    # union busy time (any queue)
    ev = sorted((s, e) for ks in per_q.values() for s, e, _ in ks)
    cur_s, cur_e, union = ev[0][0], ev[0][1], 0
    for s, e in ev[1:]:
        if s > cur_e:
            union += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    union += cur_e - cur_s
    print('GPU busy (union over queues) %.3f ms of %.3f ms span (%.1f %%)' % (union / 1e6, (t1 - t0) / 1e6, 100.0 * union / (t1 - t0)))
    agg = collections.defaultdict(lambda: [0, 0])
    for ks in per_q.values():
        for s, e, n in ks:
            agg[n][0] += 1
            agg[n][1] += e - s
    for n, (c, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]:
        print('  %-48s %5d  %9.3f ms  avg %8.2f us' % (n, c, ns / 1e6, ns / 1e3 / c))


if __name__ == '__main__':
    main(sys.argv[1])
