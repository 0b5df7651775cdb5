#!/bin/bash
# round 6: rehearsal of the strips on ONE card (gloo ranks sharing the device): halos in the all-reduce vs from
# neighbour to neighbour (flow_peer), eighth-size proxy (772 x 180, 1.2 M DoF)
set -o pipefail
mkdir -p gpurun_out
for N in 2 4; do
for H in allreduce peer; do
  timeout -k 10 400 python bench.py --gpus $N --backend gloo --halos $H --nx 772 --mu 0.00565 --headline plateau --no-cpu-baseline --no-hbm-resident --no-fast-leg --steps 20 --warmup 5 > gpurun_out/r6_proxy_${N}_${H}.json 2> gpurun_out/r6_proxy_${N}_${H}.err
  echo "N=$N halos=$H rc=$?"
  python - <<PY
import json
try:
    d = json.loads(open('gpurun_out/r6_proxy_${N}_${H}.json').read().strip().splitlines()[-1])
    c = d['config']
    print('  %.2f ms/step, collectives/step %s, launches/step %.0f, collective_us %s' % (d['ms_per_step'], c.get('collectives_per_step'), c.get('launches_per_step'), c.get('collective_us')))
except Exception as e:
    print('  no line:', e)
PY
done; done 2>&1 | tee gpurun_out/r6_proxy_ranks.txt
