#!/bin/bash
# SQ counter survey of one kernel: tools/pmc_survey.sh <tag> <script args...>; one rocprofv3 run per group
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
tag=$1; shift
G1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM"
G2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY"
G3="SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_ANY"
G4="SQ_LEVEL_WAVES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_IFETCH SQ_THREAD_CYCLES_VALU SQ_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_SMEM"
i=0
for g in "$G1" "$G2" "$G3" "$G4"; do
  i=$((i+1))
  rocprofv3 --pmc $g --kernel-trace --output-format csv -d gpurun_out/survey_${tag}_g$i -- python tools/attn_only.py "$@" > gpurun_out/survey_${tag}_g$i.log 2>&1
done
python tools/pmc_survey_summary.py $tag
