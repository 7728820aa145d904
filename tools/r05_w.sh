set -x
export HFL_PROBES=1   # the HFL_* schedule knobs below are probe switches (hotformerloc_amd/model.py)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/r05_w_ab.log
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_kernels.py -x -q -m gpu -k "attn_ws or early_phase or native_block or full_size or merged" 2>&1 | grep -v amdgpu.ids | tail -6 > gpurun_out/r05_w_test.log
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 40 --warmup 10 --no-extras --no-cpu-baseline "${CFG[@]}" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$tag', d['value'], d['ms_per_step'])" >> gpurun_out/r05_w_ab.log; }
for i in 1 2 3; do
CFG=(); run wp_default X=1
CFG=(); run wp_two_launches HFL_ATTN_WS=0
CFG=(--config oxford --batch 64); run oxford_default X=1
CFG=(--config oxford --batch 64); run oxford_two_launches HFL_ATTN_WS=0
done
