#!/bin/bash
# round-6 evidence: full GPU test log, kernel traces of the default step (overlapped + one-stream) and of the matched-precision
# step, PMC traffic of both, counter survey of hfl_linear_x6, default bench line (with the CPU baseline).
#   tools/r06_final.sh [tag]     (run on the GPU box; writes gpurun_out/<tag>_*)
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
tag=${1:-r06}
out=gpurun_out
mkdir -p $out
python -c "from hotformerloc_amd import _native; _native.load(); print('library ok')" || exit 1
python -m pytest tests -m gpu -x -q --durations=10 > $out/${tag}_gputest.log 2>&1; tail -16 $out/${tag}_gputest.log
tools/profile_round.sh $tag > $out/${tag}_profile_round.log 2>&1
tools/profile_serial.sh $tag > $out/${tag}_profile_serial.log 2>&1
# ---- the matched-precision step: kernel trace + PMC traffic
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_x6_stats -- python bench.py --gemm x6 --serial-streams --steps 16 --warmup 2 --no-extras --no-cpu-baseline > $out/${tag}_x6_stats.log 2>&1
cp "$(find $out/${tag}_x6_stats -name '*kernel_stats.csv' | head -1)" $out/${tag}_matched_serial_kernel_stats.csv
python tools/summarize_rocprof.py $out/${tag}_matched_serial_kernel_stats.csv 18 > $out/${tag}_matched_serial_summary_table.md
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/${tag}_x6_pmc_$c -- python bench.py --gemm x6 --steps 3 --warmup 1 --no-extras --no-cpu-baseline > $out/${tag}_x6_pmc_$c.log 2>&1
done
f=$(find $out/${tag}_x6_pmc_FETCH_SIZE -name '*counter_collection.csv' | head -1)
w=$(find $out/${tag}_x6_pmc_WRITE_SIZE -name '*counter_collection.csv' | head -1)
python tools/pmc_summary.py "$f" "$w" $out/${tag}_pmc_traffic_matched.json > $out/${tag}_pmc_summary_matched.txt
python - "$tag" <<'PY'
import json, sys
tag = sys.argv[1]
a = json.load(open('gpurun_out/%s_pmc_traffic.json' % tag))
b = json.load(open('gpurun_out/%s_pmc_traffic_matched.json' % tag))
for k in ('gemm_x6_kernel', 'window_attn_kernel_v4'):
    if k in b:
        a[k] = dict(b[k], step='matched-precision step (bench.py --gemm x6)')
json.dump(a, open('gpurun_out/%s_pmc_traffic.json' % tag, 'w'), indent=1, sort_keys=True)
print({k: a[k]['hbm_bytes_per_launch'] for k in ('gemm_x6_kernel', 'ln_mlp_fused_kernel', 'cpe_fwd_kernel') if k in a})
PY
# ---- counter survey of hfl_linear_x6 at the depth-4 fc1 shape (MFMA busy, waits, LDS)
bash tools/pmc_survey2.sh ${tag}_x6_fc1 gemm_x6_kernel tools/x6_one.py 68167 256 1024 0 > $out/${tag}_x6_counters.txt 2>&1
bash tools/cpe_counters.sh ${tag}
# ---- training step (config 3): kernel trace, attention-backward bench, plain and checkpointed step
bash tools/prof_train.sh ${tag}_final_train > $out/${tag}_final_train_prof.log 2>&1
rm -rf $out/${tag}_final_train_stats
timeout 600 python tools/attn_bwd_bench.py cs-wild-places 64 8192 > $out/${tag}_attn_bwd_bench_cs_wild_places.log 2>&1
for i in 1 2; do
  timeout 600 python bench.py --train --config cs-wild-places --steps 5 --warmup 3 --no-cpu-baseline --no-extras > $out/${tag}_train_cs_$i.json 2>/dev/null
  HFL_CHECKPOINT=always timeout 600 python bench.py --train --config cs-wild-places --steps 5 --warmup 3 --no-cpu-baseline --no-extras > $out/${tag}_train_cs_checkpointed_$i.json 2>/dev/null
done
rm -rf $out/${tag}_stats $out/${tag}_serial_stats $out/${tag}_pmc_FETCH_SIZE $out/${tag}_pmc_WRITE_SIZE $out/${tag}_x6_stats $out/${tag}_x6_pmc_FETCH_SIZE $out/${tag}_x6_pmc_WRITE_SIZE $out/survey_${tag}_x6_fc1_g*
head -16 $out/${tag}_matched_serial_summary_table.md
tail -c 800 $out/${tag}_bench_default.json
