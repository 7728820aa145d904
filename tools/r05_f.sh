set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -12 > gpurun_out/r05_f_gputest.log
timeout 300 python tools/host_issue_probe.py > gpurun_out/r05_f_graph_probe.log 2>&1
timeout 900 python bench.py > gpurun_out/r05_f_bench.json 2> gpurun_out/r05_f_bench.err
tail -c 300 gpurun_out/r05_f_bench.err
