set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "early_phase or native_block or full_size" 2>&1 | tail -5 > gpurun_out/r05_m_test.log
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 40 --warmup 10 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$tag', d['value'], d['ms_per_step'], d.get('parity_reference'))" >> gpurun_out/r05_m_ab.log; }
for i in 1 2 3; do
run ws0 HFL_ATTN_WS=0
run ws1 HFL_ATTN_WS=1
done
HFL_ATTN_WS=1 timeout 300 python tools/forward_phases.py > gpurun_out/r05_m_phases_ws1.log 2>&1
