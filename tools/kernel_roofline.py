"""By-kernel HBM roofline table of one step: kernel-trace durations (one-stream schedule) x PMC HBM bytes per launch.
    python tools/kernel_roofline.py <serial_kernel_stats.csv> <pmc_traffic.json> <steps in the trace> <steps in the PMC pass>
Both inputs are written by tools/profile_serial.sh / tools/profile_round.sh (rocprofv3 --kernel-trace --stats, and the separate
--pmc FETCH_SIZE / WRITE_SIZE passes corrected as MI355X_MICROARCH.md prescribes).  'frac' = bytes / time / 8 TB/s: how close
each kernel group runs to the HBM roof with the bytes it ACTUALLY moved; the algorithmic-byte rooflines of the two attention
kernels are in bench.py's JSON line (roofline, roofline_fused)."""
import collections
import csv
import json
import re
import sys

HBM_PEAK = 8000.0  # GB/s, MI355X_MICROARCH.md


def group(name):
    m = re.search(r'\(anonymous namespace\)::(\w+)', name)
    if m:
        return m.group(1)
    if 'Cijk_' in name:
        return 'hipblaslt_gemm'
    if 'CatArrayBatchedCopy_contig' in name:
        return 'CatArrayBatchedCopy_contig'
    return 'other'


def main():
    stats, pmc = sys.argv[1], json.load(open(sys.argv[2]))
    steps, pmc_steps = float(sys.argv[3]), float(sys.argv[4])
    t = collections.defaultdict(lambda: [0, 0])
    for r in csv.DictReader(open(stats)):
        g = group(r['Name'])
        t[g][0] += int(r['Calls'])
        t[g][1] += int(r['TotalDurationNs'])
    rows = []
    for g, (calls, ns) in t.items():
        p = pmc.get(g)
        if not p or g == 'other':
            continue
        per_step_bytes = p['hbm_bytes_per_launch'] * p['launches'] / pmc_steps
        ms = ns / steps / 1e6
        rows.append((ms, g, calls / steps, ns / calls / 1e3, p['hbm_bytes_per_launch'] / 1e6, per_step_bytes / 1e9,
                     per_step_bytes / (ms * 1e-3) / 1e9))
    rows.sort(reverse=True)
    tot_ms = sum(r[0] for r in rows)
    tot_gb = sum(r[5] for r in rows)
    print('| kernel group | launches/step | avg us | ms/step | HBM MB/launch (PMC) | HBM GB/step | GB/s | frac of 8 TB/s |')
    print('|---|---|---|---|---|---|---|---|')
    for ms, g, n, us, mb, gb, gbps in rows:
        print('| %s | %.1f | %.1f | %.3f | %.1f | %.2f | %.0f | %.2f |' % (g, n, us, ms, mb, gb, gbps, gbps / HBM_PEAK))
    print('| **all listed** | | | %.3f | | %.2f | %.0f | %.2f |' % (tot_ms, tot_gb, tot_gb / tot_ms * 1e3,
                                                                  tot_gb / tot_ms * 1e3 / HBM_PEAK))


if __name__ == '__main__':
    main()
