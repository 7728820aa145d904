#!/bin/bash
# L1 / TA / L2 counters of cpe_fwd_kernel at the bench's depth-4 shape: one rocprofv3 --pmc pass per
# small counter group (kernel-trace only), counters picked from what `rocprofv3 -L` offers on this box.
#   tools/cpe_counters.sh <tag>
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
tag=$1
out=gpurun_out
rocprofv3 -L > $out/${tag}_counter_list.txt 2>&1
want="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_TOTAL_READ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TA_TA_BUSY_sum TA_BUSY_avr TA_BUFFER_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TD_LOAD_WAVEFRONT_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUSY_sum TCC_TAG_STALL_sum GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM FETCH_SIZE WRITE_SIZE L2CacheHit"
have=()
for c in $want; do
  if grep -qw "$c" $out/${tag}_counter_list.txt; then have+=("$c"); fi
done
echo "counters present: ${have[*]}" > $out/${tag}_cpe_counters.txt
i=0
n=${#have[@]}
while [ $i -lt $n ]; do
  g="${have[$i]}"
  if [ $((i+1)) -lt $n ]; then g="$g ${have[$((i+1))]}"; fi
  rocprofv3 --pmc $g --kernel-trace --output-format csv -d $out/${tag}_cpe_g$i -- python tools/cpe_only.py > $out/${tag}_cpe_g$i.log 2>&1
  i=$((i+2))
done
python - "$tag" >> $out/${tag}_cpe_counters.txt <<'PY'
import csv, glob, collections, sys
tag = sys.argv[1]
agg = collections.defaultdict(lambda: [0, 0.0])
dur = []
for f in glob.glob('gpurun_out/%s_cpe_g*/**/*counter_collection.csv' % tag, recursive=True):
    for r in csv.DictReader(open(f)):
        if 'cpe_fwd_kernel' not in r['Kernel_Name']:
            continue
        a = agg[r['Counter_Name']]; a[0] += 1; a[1] += float(r['Counter_Value'])
for f in glob.glob('gpurun_out/%s_cpe_g*/**/*kernel_trace.csv' % tag, recursive=True):
    for r in csv.DictReader(open(f)):
        if 'cpe_fwd_kernel' in r['Kernel_Name']:
            dur.append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
print('cpe_fwd_kernel<64>, depth 4 of the bench workload (66 775 rows x 256 channels), per launch:')
for k in sorted(agg):
    n, v = agg[k]
    print('%-40s %18.0f  (%d samples)' % (k, v / n, n))
if dur:
    dur.sort(); print('kernel duration under the profiler: median %.1f us over %d launches' % (dur[len(dur) // 2] / 1e3, len(dur)))
PY
rm -rf $out/${tag}_cpe_g*
