#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_tlilu.py tests/test_pmg.py tests/test_hip_ilu.py tests/test_hip_heat.py -m gpu -x -q -s > gpurun_out/r6_tl_tests.log 2>&1
echo "tests rc $?" | tee -a gpurun_out/r6_tl_tests.log
tail -5 gpurun_out/r6_tl_tests.log
for v in "tlilu 0,1,1" "tlilu 0,1,2" "tlilu 1,1,1"; do
  set -- $v
  FALLBACK=$1 TLILU=$2 timeout -k 10 500 python tools/graded_mesh_time.py 3.2e-4 20 2>&1 | grep -v "^generated\|amdgpu.ids" | tee -a gpurun_out/r6_graded_tl2.txt
done
timeout -k 10 600 python tools/boussinesq_time.py 400 15 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r6_bq_tl.txt
tail -12 gpurun_out/r6_bq_tl.txt
