#!/bin/bash
# the proxy with the graphs of ONE loop at a time (1 CG, 2 GMRES, 4 mass solver)
set -e
out=gpurun_out/ab_graphs
mkdir -p $out
for rep in 1 2; do
for sites in ${SITES:-0 1 2 4}; do
  if [ $sites = 0 ]; then export FLOW_AMD_GRAPHS=0; else export FLOW_AMD_GRAPHS=1 FLOW_AMD_GRAPH_SITES=$sites; fi
  timeout -k 10 300 python3 bench.py --nx ${NX:-772} --mu ${MU:-0.00565} --no-cpu-baseline --developed 0 > $out/sites_${sites}_$rep.json 2> $out/sites_${sites}_$rep.err
done
done
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/ab_graphs/sites_*.json')):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    c = d['config']
    print('%-28s %.2f steps/s %.3f ms  launches/step %s' % (
        f.split('/')[-1], d['value'], d['ms_per_step'], c.get('launches_per_step')))
PY
