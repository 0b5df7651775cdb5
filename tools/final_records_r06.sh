#!/bin/bash
# Round 6's records (GPU box): the bench lines, the 4000-step run, config 4, the mid-size channels, the profiles.
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/final_records.sh r06 > gpurun_out/final_r06_summary.txt 2>&1; echo "final_records rc $?"
tail -12 gpurun_out/final_r06_summary.txt
timeout -k 10 500 python tools/long_run.py 4000 > gpurun_out/long_run_r06_4000.txt 2>&1; tail -2 gpurun_out/long_run_r06_4000.txt
timeout -k 10 600 python tools/boussinesq_time.py 400 18 2>&1 | grep -v amdgpu.ids > gpurun_out/boussinesq_c4_r06.txt; grep "^step 1[2-4]" gpurun_out/boussinesq_c4_r06.txt | cut -c1-200
timeout -k 10 500 python tools/graded_mesh_time.py 3.2e-4 20 2>&1 | grep -v "amdgpu.ids" > gpurun_out/graded_mesh_r06.txt; cat gpurun_out/graded_mesh_r06.txt
bash profiles/run_profiles.sh r06 > gpurun_out/run_profiles_r06.log 2>&1; echo "profiles rc $?"; tail -3 gpurun_out/run_profiles_r06.log
