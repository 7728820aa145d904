set -x
export HFL_PROBES=1   # the HFL_* schedule knobs below are probe switches (hotformerloc_amd/model.py)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "x6" > gpurun_out/r06_f_x6_tests.log 2>&1
tail -5 gpurun_out/r06_f_x6_tests.log
HFL_EXTRA_HIPCC_FLAGS=-DHFL_X6_STAMPS python -m hotformerloc_amd.build --force > gpurun_out/r06_f_build.log 2>&1
tail -2 gpurun_out/r06_f_build.log
for sh in 4 14; do
  timeout 300 python tools/x6_stamps.py 68167 256 1024 $sh > gpurun_out/r06_f_stamps_fc1_$sh.log 2>&1
  cat gpurun_out/r06_f_stamps_fc1_$sh.log
done
