#!/bin/bash
# A/B of an environment switch on the DEFAULT bench line (plateau + developed
# street windows), without / with "$1", alternating, twice.
set -e
out=gpurun_out/ab_envd
rm -rf $out; mkdir -p $out
for rep in 1 2; do
  for v in a b; do
    if [ $v = b ]; then export "$1"; fi
    timeout -k 10 500 python3 bench.py --no-cpu-baseline > $out/bench_${v}_$rep.json 2> $out/bench_${v}_$rep.err
    if [ $v = b ]; then unset "${1%%=*}"; fi
  done
done
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/ab_envd/*.json')):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print('%-20s plateau %.2f steps/s %.3f ms   developed %.2f steps/s %.3f ms  roofline %.3f' % (
        f.split('/')[-1], d['value'], d['ms_per_step'], d.get('value_developed', 0),
        d.get('ms_per_step_developed', 0), d['roofline']['frac']))
PY
