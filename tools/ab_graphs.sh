#!/bin/bash
# A/B of the HIP-graph replay of the iteration bodies (csrc/graph_replay.hip) on
# the GPU box: the eighth-size proxy without (FLOW_AMD_GRAPHS=0) and with the
# graphs, twice each, alternating; then the headline's plateau window without
# and with the graphs FORCED on (at that size the default leaves them off).
set -e
out=gpurun_out/ab_graphs
mkdir -p $out
for rep in 1 2; do
  for g in 0 1; do
    FLOW_AMD_GRAPHS=$g timeout -k 10 300 python3 bench.py --nx 772 --mu 0.00565 --no-cpu-baseline --developed 0 > $out/proxy_${g}_$rep.json 2> $out/proxy_${g}_$rep.err
  done
done
for g in 0 1; do
  FLOW_AMD_GRAPHS=$g timeout -k 10 400 python3 bench.py --no-cpu-baseline --developed 0 --steps 20 --warmup 5 > $out/head_$g.json 2> $out/head_$g.err
done
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/ab_graphs/*.json')):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    c = d['config']
    print('%-28s %.2f steps/s %.3f ms  launches/step %s  apps %.1f  p %.1f' % (
        f.split('/')[-1], d['value'], d['ms_per_step'], c.get('launches_per_step'),
        sum(c['newton_linear_applications']) / float(len(c['newton_linear_applications'])),
        sum(c['pressure_cg_iterations']) / float(len(c['pressure_cg_iterations']))))
PY
