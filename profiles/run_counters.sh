#!/bin/bash
# Counter passes over the bench's timed window (GPU box, through gpurun, from
# the repo root):   bash profiles/run_counters.sh rNN [bench args]
# Separate rocprofv3 --pmc runs of the SAME bench command (no trace domains
# beside them): FETCH_SIZE, WRITE_SIZE, and two SQ sets.  The per-kernel table
# (medians over each kernel's dispatches, by dispatch size) lands in
# gpurun_out/counters_rNN.md; copy it into profiles/.
set -e
TAG=${1:-r04}
shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_$TAG
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 4 --warmup 2 --developed 0 --no-cpu-baseline --no-hbm-resident --no-fast-leg --spmv-reps 5 $@"
i=0
for SET in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --output-format csv -d $OUT/pass$i -- python3 $R/bench.py $ARGS > $OUT/bench_pass$i.json 2> $OUT/pass$i.err
  echo "pass $i done: $SET"
done
python3 $R/profiles/summarize.py counters $R/gpurun_out/counters_$TAG.md $OUT/pass1 $OUT/pass2 $OUT/pass3 $OUT/pass4 > /dev/null
rm -rf $OUT/pass1 $OUT/pass2 $OUT/pass3 $OUT/pass4
