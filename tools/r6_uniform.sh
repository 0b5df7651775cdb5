#!/bin/bash
# round 6: the reference's own mesh (one lcar) at its own size and at ~1 M DoF
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_loaded_meshes.py -m gpu -x -q > gpurun_out/r6_uniform_tests.log 2>&1
echo "tests rc $?"; tail -3 gpurun_out/r6_uniform_tests.log
{
FAR=1 NO_STRUCTURED=1 timeout -k 10 300 python tools/graded_mesh_time.py 5e-3 200 2>&1 | grep -v "amdgpu.ids"
FLOW_AMD_GRAPHS=1 FAR=1 NO_STRUCTURED=1 timeout -k 10 300 python tools/graded_mesh_time.py 5e-3 200 2>&1 | grep -v "amdgpu.ids" | sed 's/^/FLOW_AMD_GRAPHS=1: /'
FAR=1 NO_STRUCTURED=1 timeout -k 10 500 python tools/graded_mesh_time.py 9.4e-4 20 2>&1 | grep -v "amdgpu.ids"
} | tee gpurun_out/r6_uniform.txt
