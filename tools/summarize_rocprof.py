"""Summarise a rocprofv3 --kernel-trace --stats kernel_stats.csv into a short table
(grouped: hipBLASLt GEMMs, torch elementwise/LayerNorm, this repo's HIP kernels)."""
import csv, sys, re, collections

def group(name):
    if name.startswith('Cijk_'): return 'hipBLASLt fp32 GEMM (torch Linear/mm/bmm)'
    m = re.search(r'\(anonymous namespace\)::(\w+)', name)
    if m and not name.startswith('void at::') and 'at::native' not in name:
        t = re.search(r'<([^>]*)>', name)
        return 'hfl:' + m.group(1) + (('<%s>' % t.group(1)) if t else '')
    if 'layer_norm' in name: return 'torch LayerNorm'
    if 'Gelu' in name: return 'torch GELU'
    if 'CatArray' in name: return 'torch cat'
    if 'elementwise' in name or 'copyBuffer' in name or 'gather_kernel' in name or 'index' in name: return 'torch elementwise/copy/index'
    return 'other: ' + name[:60]

def main(path, steps):
    rows = list(csv.DictReader(open(path)))
    agg = collections.OrderedDict()
    total = 0
    for r in rows:
        g = group(r['Name']); ns = int(r['TotalDurationNs']); total += ns
        a = agg.setdefault(g, [0, 0]); a[0] += int(r['Calls']); a[1] += ns
    print('| kernel group | calls | total ms | ms/step (%d steps+warmup) | avg us | %% |' % steps)
    print('|---|---|---|---|---|---|')
    for g, (c, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print('| %s | %d | %.3f | %.3f | %.2f | %.1f |' % (g, c, ns/1e6, ns/1e6/steps, ns/1e3/c, 100.0*ns/total))
    print('| TOTAL | | %.3f | %.3f | | 100 |' % (total/1e6, total/1e6/steps))

if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 1)
