#!/bin/bash
# LDS / MFMA / issue counters of attn_ws_kernel at the bench's depth-4 shape: one rocprofv3 --pmc pass per counter pair.
#   tools/attn_ws_counters.sh <tag>
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
export HFL_WS_ONLY=1
tag=$1
out=gpurun_out
mkdir -p $out
groups=("SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY" "SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL" "SQ_INSTS_LDS_LOAD_BANDWIDTH SQ_INSTS_LDS_STORE_BANDWIDTH" "SQ_LDS_CMD_FIFO_FULL SQ_INST_LEVEL_LDS" "SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_INSTS_MFMA")
i=0
for g in "${groups[@]}"; do
  rocprofv3 --pmc $g --kernel-trace --output-format csv -d $out/${tag}_ws_g$i -- python tools/attn_ws_probe.py > $out/${tag}_ws_g$i.log 2>&1
  i=$((i+1))
done
python - "$tag" > $out/${tag}_attn_ws_counters.txt <<'PY'
import csv, glob, collections, sys
tag = sys.argv[1]
agg = collections.defaultdict(lambda: [0, 0.0])
dur = []
for f in glob.glob('gpurun_out/%s_ws_g*/**/*counter_collection.csv' % tag, recursive=True):
    for r in csv.DictReader(open(f)):
        if 'attn_ws_kernel' not in r['Kernel_Name']:
            continue
        a = agg[r['Counter_Name']]; a[0] += 1; a[1] += float(r['Counter_Value'])
for f in glob.glob('gpurun_out/%s_ws_g*/**/*kernel_trace.csv' % tag, recursive=True):
    for r in csv.DictReader(open(f)):
        if 'attn_ws_kernel' in r['Kernel_Name']:
            dur.append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
print('attn_ws_kernel, depth 4 of the bench workload, per launch:')
for k in sorted(agg):
    n, v = agg[k]
    print('%-40s %18.0f  (%d samples)' % (k, v / n, n))
if dur:
    dur.sort(); print('kernel duration under the profiler: median %.1f us over %d launches' % (dur[len(dur) // 2] / 1e3, len(dur)))
PY
rm -rf $out/${tag}_ws_g*
