#!/bin/bash
# round 6: start-vector histories fed by the steps' TOTAL Newton increments (+ gated adaptive forcing), 4000-step runs
set -o pipefail
mkdir -p gpurun_out
N=${N:-4000}
{
echo "== defaults"; timeout -k 10 400 python tools/long_run.py $N 2>&1 | tail -2
echo "== linear_history=total"; timeout -k 10 400 python tools/long_run.py $N newton.linear_history=total 2>&1 | tail -2
echo "== linear_history=total adaptive_forcing"; timeout -k 10 400 python tools/long_run.py $N newton.linear_history=total newton.adaptive_forcing=1 2>&1 | tail -2
echo "== adaptive_forcing (round 5's)"; timeout -k 10 400 python tools/long_run.py $N newton.adaptive_forcing=1 2>&1 | tail -2
} | tee gpurun_out/r6_history.txt
