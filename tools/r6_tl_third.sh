#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for sw in 1 3 6; do
FLOW_AMD_TL_PROBE_SWEEPS=$sw FALLBACK=tlilu TLILU=0,1,1 timeout -k 10 500 python tools/graded_mesh_time.py 3.2e-4 4 2>&1 | grep "tl probe" | sort | uniq -c | tee -a gpurun_out/r6_probe.txt
done
