#!/bin/bash
# Run on the GPU box (through gpurun) from the repo root:
#   bash profiles/run_profiles.sh rNN
# Three separate rocprofv3 runs of the SAME bench command: kernel trace (+stats),
# FETCH_SIZE pass, WRITE_SIZE pass.  Summaries land in gpurun_out/ and are then
# copied into profiles/ (tracked).
set -e
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 10 --warmup 20 --developed 0 --no-cpu-baseline --no-hbm-resident --no-fast-leg --spmv-reps 50"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err
echo trace done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/bench.py $ARGS > $OUT/bench_fetch.json 2> $OUT/fetch.err
echo fetch done
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/bench.py $ARGS > $OUT/bench_write.json 2> $OUT/write.err
echo write done
python3 $R/profiles/summarize.py stats $OUT/trace $R/gpurun_out/kernel_stats_$TAG.md
python3 $R/profiles/summarize.py gaps $OUT/trace $R/gpurun_out/gaps_$TAG.md
# the timed steps only (kernel budget of one step): between the marker launches
# bench.py puts around them; fails unless exactly 10 steps lie there
python3 $R/profiles/summarize.py stats $OUT/trace $R/gpurun_out/kernel_stats_${TAG}_timed_steps.md timed:10 > /dev/null
python3 $R/profiles/summarize.py gaps $OUT/trace $R/gpurun_out/gaps_${TAG}_timed_steps.md timed:10 > /dev/null
# warm replay (spmv_stream_kernel<false>, back-to-back) and the in-solver launches
# (spmv_stream_kernel<true>) on the pressure matrix, picked by its dispatch size
GRID=$(python3 -c "import json; print(json.load(open('$OUT/bench_trace.json'))['config']['pressure_spmv_grid'])")
python3 $R/profiles/summarize.py pmc $OUT/fetch $OUT/write $R/gpurun_out/spmv_traffic_$TAG.json "flow::spmv_stream_kernel<false>" $GRID
python3 $R/profiles/summarize.py pmc $OUT/fetch $OUT/write $R/gpurun_out/spmv_traffic_${TAG}_in_solver.json "flow::spmv_stream_kernel<true>" $GRID
# keep the merge-back small: drop the raw traces
rm -rf $OUT/trace $OUT/fetch $OUT/write
