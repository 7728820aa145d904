#!/bin/bash
# SQ / TCC counter survey of one kernel: tools/pmc_survey2.sh <tag> <kernel name substring> <script> [args...]
# one rocprofv3 run per counter group (the program itself after --, no wrapper)
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
tag=$1; kern=$2; shift; shift
G1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM"
G2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY"
G3="SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_ANY"
G4="SQ_LEVEL_WAVES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_IFETCH SQ_THREAD_CYCLES_VALU SQ_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_SMEM"
G5="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE"
G6="FETCH_SIZE"
G7="WRITE_SIZE"
i=0
for g in "$G1" "$G2" "$G3" "$G4" "$G5" "$G6" "$G7"; do
  i=$((i+1))
  rocprofv3 --pmc $g --kernel-trace --output-format csv -d gpurun_out/survey_${tag}_g$i -- python "$@" > gpurun_out/survey_${tag}_g$i.log 2>&1
done
python - "$tag" "$kern" <<'PY'
import csv, glob, collections, sys
tag, kern = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: [0, 0.0])
dur = []
for f in glob.glob('gpurun_out/survey_%s_g*/**/*counter_collection.csv' % tag, recursive=True):
    for r in csv.DictReader(open(f)):
        if kern not in r['Kernel_Name']:
            continue
        a = agg[r['Counter_Name']]; a[0] += 1; a[1] += float(r['Counter_Value'])
        if 'Start_Timestamp' in r and 'End_Timestamp' in r:
            dur.append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for k in sorted(agg):
    n, v = agg[k]
    print('%-32s %16.0f  (%d samples)' % (k, v / n, n))
if dur:
    dur.sort(); print('kernel duration under the profiler: median %.1f us' % (dur[len(dur) // 2] / 1e3))
PY
