#!/bin/bash
# A/B of an environment switch ("$1", e.g. HSA_ENABLE_INTERRUPT=0) on the GPU
# box: proxy and headline plateau windows without / with it, twice, alternating.
set -e
out=gpurun_out/ab_envq
rm -rf $out; mkdir -p $out
for rep in 1 2; do
  for v in ${ORDER:-a b}; do
    if [ $v = b ]; then export "$1"; fi
    timeout -k 10 300 python3 bench.py --nx 772 --mu 0.00565 --no-cpu-baseline --developed 0 --steps 100 --warmup 40 > $out/proxy_${v}_$rep.json 2> $out/proxy_${v}_$rep.err
    timeout -k 10 400 python3 bench.py --no-cpu-baseline --developed 0 --steps 40 --warmup 20 > $out/head_${v}_$rep.json 2> $out/head_${v}_$rep.err
    if [ $v = b ]; then unset "${1%%=*}"; fi
  done
done
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/ab_envq/*.json')):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print('%-20s %.2f steps/s %.3f ms' % (f.split('/')[-1], d['value'], d['ms_per_step']))
PY
