#!/bin/bash
# as ab_graphs.sh with long windows (the captures behind a rebuild of the lagged
# Newton preconditioner amortised): NX / MU / STEPS from the environment
set -e
out=gpurun_out/ab_graphs_long
mkdir -p $out
for rep in 1 2; do
  for g in 0 1; do
    FLOW_AMD_GRAPHS=$g timeout -k 10 300 python3 bench.py --nx ${NX:-772} --mu ${MU:-0.00565} --no-cpu-baseline --developed 0 --steps ${STEPS:-150} --warmup 60 > $out/nx${NX:-772}_${g}_$rep.json 2> $out/nx${NX:-772}_${g}_$rep.err
  done
done
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/ab_graphs_long/*.json')):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    c = d['config']
    print('%-22s %.2f steps/s %.3f ms  launches/step %.1f  %s' % (
        f.split('/')[-1], d['value'], d['ms_per_step'], c.get('launches_per_step'),
        c.get('graph_replay')))
PY
