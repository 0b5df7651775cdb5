#!/bin/bash
# The bench lines recorded under profiles/ at the end of a round (GPU box):
#   tools/final_records.sh rNN
set -e
TAG=${1:-r04}
cd $GRAFT_REPO_ROOT
O=gpurun_out/final_$TAG
mkdir -p $O
python3 bench.py > $O/bench.json 2> $O/bench.err
echo default done
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/driver_window.json 2> $O/driver_window.err
python3 bench.py --nx 772 --mu 0.00565 --no-cpu-baseline > $O/proxy.json 2> $O/proxy.err
python3 bench.py --nx 772 --mu 0.00565 --no-cpu-baseline --shard-single > $O/proxy_sharded.json 2> $O/proxy_sharded.err
FLOW_AMD_RCCL_DIRECT=1 python3 bench.py --nx 772 --mu 0.00565 --no-cpu-baseline --shard-single > $O/proxy_sharded_direct.json 2> $O/proxy_sharded_direct.err
python3 bench.py --nx 1196 --velocity-degree 1 --no-cpu-baseline > $O/c2.json 2> $O/c2.err
echo benches done
python3 tools/long_run.py 500 > $O/long_run.txt 2>&1
python3 - $O <<'PY'
import json, sys, os
o = sys.argv[1]
for name in ('bench', 'driver_window', 'proxy', 'proxy_sharded', 'proxy_sharded_direct', 'c2'):
    d = json.loads(open(os.path.join(o, name + '.json')).read().strip().splitlines()[-1])
    c = d['config']
    ap = c.get('newton_linear_applications', [0])
    print(name, '%.2f steps/s %.3f ms' % (d['value'], d['ms_per_step']),
          'apps %.2f' % (sum(ap) / float(len(ap))),
          'roofline %.3f' % d['roofline']['frac'] if d.get('roofline') else '',
          c.get('collectives_per_step'),
          'developed %.2f (%.3f ms), period %.2f (%.3f ms)' % (
              d['value_developed'], d['ms_per_step_developed'],
              d.get('value_developed_period', 0),
              d.get('ms_per_step_developed_period', 0))
          if 'value_developed' in d else '')
PY
tail -3 $O/long_run.txt
