#!/bin/bash
# quick kernel-time table of a short bench window (GPU box, via gpurun):
#   bash tools/prof_quick.sh [extra bench args]
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_quick
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 10 --warmup 20 --developed 0 --no-cpu-baseline --no-hbm-resident --no-fast-leg --spmv-reps 5 "$@" > $OUT/bench.json 2> $OUT/err.txt
python3 $R/profiles/summarize.py stats $OUT/trace $R/gpurun_out/quick_stats.md timed:10 > /dev/null
python3 $R/profiles/summarize.py gaps $OUT/trace $R/gpurun_out/quick_gaps.md timed:10 > /dev/null
python3 $R/profiles/summarize.py sequence $OUT/trace $R/gpurun_out/quick_sequence.txt timed
rm -rf $OUT/trace
