#!/bin/bash
# like ab_env.sh, headline only, a long window (60 steps)
set -e
out=gpurun_out/ab
mkdir -p $out
run() {
  timeout -k 10 500 python3 bench.py --no-cpu-baseline --no-hbm-resident --no-fast-leg --steps 60 --warmup 5 > $out/long_$1.json 2> $out/long_$1.err
}
run a
export "$1"
run b
python3 - <<'PY'
import json
for v in ('a', 'b'):
    d = json.loads(open('gpurun_out/ab/long_%s.json' % v).read().strip().splitlines()[-1])
    c = d['config']
    print(v, '%.2f steps/s %.3f ms' % (d['value'], d['ms_per_step']),
          'apps', sum(c['newton_linear_applications']) / float(len(c['newton_linear_applications'])),
          'p', sum(c['pressure_cg_iterations']) / float(len(c['pressure_cg_iterations'])), c['newton_linear_applications'])
PY
