#!/bin/bash
# A/B of a build flag on the GPU box: bench lines (proxy + headline) of the
# library as built, then rebuilt with EXTRA="$1".  Usage: tools/ab_bench.sh -DFLAG
set -e
out=gpurun_out/ab
mkdir -p $out
run() {
  timeout -k 10 300 python3 bench.py --nx 772 --mu 0.00565 --no-cpu-baseline > $out/proxy_$1.json 2> $out/proxy_$1.err
  timeout -k 10 400 python3 bench.py --no-cpu-baseline --steps 20 --warmup 5 > $out/head_$1.json 2> $out/head_$1.err
}
run a
touch flow_amd/csrc/*.hip
make -C flow_amd/csrc -j16 EXTRA="$1" > $out/make.log 2>&1
run b
python3 - <<'PY'
import json
for w in ('proxy', 'head'):
    for v in ('a', 'b'):
        d = json.loads(open('gpurun_out/ab/%s_%s.json' % (w, v)).read().strip().splitlines()[-1])
        c = d['config']
        print(w, v, '%.2f steps/s %.3f ms' % (d['value'], d['ms_per_step']),
              'apps', sum(c['newton_linear_applications']) / float(len(c['newton_linear_applications'])),
              'p', sum(c['pressure_cg_iterations']) / float(len(c['pressure_cg_iterations'])))
PY
