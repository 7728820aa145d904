#!/bin/bash
# round-5 evidence, part B: SQ / TCC counter surveys of the window kernel, the fused attention kernel and the fused MLP (each
# stamped with its sources: bench.py reports a figure only while they are unchanged), training profile, smoke
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
out=gpurun_out
mkdir -p $out
python -c "from hotformerloc_amd import _native; _native.load(); print('library ok')" || exit 1
tools/pmc_survey2.sh r05_attn window_attn_kernel_v5 tools/attn_v5_one.py 4 4 > $out/r05_attn_counters.txt 2>&1
tools/pmc_survey2.sh r05_fused attn_fused_kernel tools/attn_fused_probe.py > $out/r05_fused_counters.txt 2>&1
tools/pmc_survey2.sh r05_mlp ln_mlp_fused_kernel tools/mlp_fused_one.py 65536 256 > $out/r05_mlp_counters.txt 2>&1
rm -rf $out/survey_r05_attn_g* $out/survey_r05_fused_g* $out/survey_r05_mlp_g*
python tools/stamp_sources.py hotformerloc_amd/csrc/attention.hip >> $out/r05_attn_counters.txt
tools/attn_ws_counters.sh r05 > $out/r05_ws_counters.log 2>&1
mv $out/r05_attn_ws_counters.txt $out/r05_ws_counters.txt
python tools/stamp_sources.py hotformerloc_amd/csrc/attn_ws.hip >> $out/r05_ws_counters.txt
python tools/stamp_sources.py hotformerloc_amd/csrc/attn_fused.hip >> $out/r05_fused_counters.txt
python tools/stamp_sources.py hotformerloc_amd/csrc/mlp_fused.hip >> $out/r05_mlp_counters.txt
tools/prof_train.sh r05_train_cs --config cs-wild-places > $out/r05_train_prof.log 2>&1
rm -rf $out/r05_train_cs_stats
timeout 900 python tools/train_ops_profile.py > $out/r05_train_ops.txt 2> $out/r05_train_ops.err
HFL_CHECKPOINT=always timeout 900 python tools/train_ops_profile.py > $out/r05_train_ops_checkpointed.txt 2>> $out/r05_train_ops.err
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/r05_smoke.log 2>&1
tail -5 $out/r05_attn_counters.txt $out/r05_fused_counters.txt $out/r05_mlp_counters.txt
head -30 $out/r05_train_cs_summary_table.md
tail -3 $out/r05_smoke.log
