cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
s=$(date +%s)
python bench.py > gpurun_out/r05_bench_timed.json 2> gpurun_out/r05_bench_timed.err
e=$(date +%s)
echo "default bench wall: $((e-s)) s" > gpurun_out/r05_bench_timed.txt
grep -E "^\[bench" gpurun_out/r05_bench_timed.err | tail -40 >> gpurun_out/r05_bench_timed.txt
