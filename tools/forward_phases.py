"""Wall-clock phases of the last traced forward of the default bench step, from a rocprofv3 --kernel-trace CSV (all queues):
per-batch tables + stem -> depth-5 OctFormer stage -> pyramid init -> each H-OSA iteration -> pooling head, with, per phase, the
busy time of the main queue and the kernels that took the most of it.
    python tools/forward_phases.py <kernel_trace.csv> [phase name substring: dump that phase kernel by kernel]"""
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
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    # a forward starts with the per-batch tap tables (tap_count_kernel, > 1 ms after the previous one) and runs up to the next
    # forward's start: the last COMPLETE forward of the trace
    starts, last = [], -10 ** 18
    for i, r in enumerate(rows):
        if 'tap_count_kernel' in r['Kernel_Name']:
            t = int(r['Start_Timestamp'])
            if t - last > 1000000:
                starts.append(i)
            last = t
    fwd = rows[starts[-2]:starts[-1]]
    # (kernels of the previous forward's head may still trail on other queues: drop what ends before the first table kernel)
    t0 = int(fwd[0]['Start_Timestamp'])
    t1 = max(int(r['End_Timestamp']) for r in fwd)
    print('last forward: %d kernels, %.3f ms wall' % (len(fwd), (t1 - t0) / 1e6))

    def first(pred, after=0):
        for r in fwd:
            if int(r['Start_Timestamp']) >= after and pred(r['Kernel_Name']):
                return int(r['Start_Timestamp'])
        return None

    marks = [('tables + stem', t0)]
    t_octf = first(lambda n: 'cpe_fwd_kernel<32>' in n)
    if t_octf:
        marks.append(('octf stage (depth 5, C=128)', t_octf))
    t_init = first(lambda n: 'relay_init' in n)
    if t_init:
        marks.append(('pyramid init (downsamples, relay tokens)', t_init))
    ra = [int(r['Start_Timestamp']) for r in fwd if 'relay_attn' in r['Kernel_Name']]
    # an iteration starts with the finest level's CPE issued BEFORE its RTSA: take the last cpe_fwd_kernel<64> start before each relay_attn
    prev = t_init or t0
    for i, t in enumerate(ra):
        cands = [int(r['Start_Timestamp']) for r in fwd if 'cpe_fwd_kernel<64>' in r['Kernel_Name'] and prev < int(r['Start_Timestamp']) < t]
        marks.append(('H-OSA iteration %d' % i, min(cands) if cands else t))
        prev = t
    t_head = first(lambda n: 'attn_pool_kernel' in n or 'segment_softmax' in n)
    marks.append(('pooling head', t_head))
    marks.append(('end', t1))
    main_q = collections.Counter(r['Queue_Id'] for r in fwd).most_common(1)[0][0]
    for (name, a), (_, b) in zip(marks[:-1], marks[1:]):
        ks = [r for r in fwd if a <= int(r['Start_Timestamp']) < b]
        busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in ks if r['Queue_Id'] == main_q)
        tot = collections.Counter()
        for r in ks:
            tot[short(r['Kernel_Name'])] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
        top = ', '.join('%s %.0f' % (k, v / 1e3) for k, v in tot.most_common(6))
        print('%-44s %8.1f us wall  main queue busy %7.1f us  %3d kernels | %s' % (name, (b - a) / 1e3, busy / 1e3, len(ks), top))
        if len(sys.argv) > 2 and sys.argv[2] in name:          # kernel by kernel: start offset, duration, queue, grid
            qs = {}
            for r in ks:
                q = qs.setdefault(r['Queue_Id'], len(qs))
                print('    +%8.1f us %7.1f us  q%d  grid %-8s wg %-5s %s' % (
                    (int(r['Start_Timestamp']) - a) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, q,
                    r.get('Grid_Size_X', r.get('Grid_Size', '?')), r.get('Workgroup_Size_X', r.get('Workgroup_Size', '?')),
                    short(r['Kernel_Name'])))


if __name__ == '__main__':
    main()
