set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/pmc_survey2.sh x6_fc1_mt4 gemm_x6_kernel tools/x6_one.py 68167 256 1024 4 > gpurun_out/r06_c_x6_fc1_mt4_counters.txt 2>&1
bash tools/pmc_survey2.sh x6_fc1_mt1 gemm_x6_kernel tools/x6_one.py 68167 256 1024 1 > gpurun_out/r06_c_x6_fc1_mt1_counters.txt 2>&1
bash tools/pmc_survey2.sh x6_fc2_mt4 gemm_x6_kernel tools/x6_one.py 68167 1024 256 4 > gpurun_out/r06_c_x6_fc2_mt4_counters.txt 2>&1
rm -rf gpurun_out/survey_x6_*
cat gpurun_out/r06_c_x6_fc1_mt4_counters.txt
