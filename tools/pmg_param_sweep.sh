#!/bin/bash
# The p-multigrid cycle's parameters on the headline workload: every argument is
# a comma-separated list of solver_parameters['newton']['pmg'] entries, e.g.
#   tools/pmg_param_sweep.sh pre=1,post=2,coarse_steps=6,ratio_coarse=16 ...
# -> ms/step and GMRES applications per step over a 40-step window each.
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmg_sweep
for spec in "$@"; do
  args=""
  for kv in ${spec//,/ }; do args="$args --newton pmg.$kv"; done
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-hbm-resident --no-fast-leg --steps 40 --warmup 5 \
    $args > gpurun_out/pmg_sweep/$spec.json 2> gpurun_out/pmg_sweep/$spec.err
  python3 - "$spec" <<'PY'
import json, sys
d = json.loads(open('gpurun_out/pmg_sweep/%s.json' % sys.argv[1]).read().strip().splitlines()[-1])
ap = d['config']['newton_linear_applications']
print(sys.argv[1], '%.3f ms/step' % d['ms_per_step'], 'apps %.2f' % (sum(ap) / float(len(ap))))
PY
done
