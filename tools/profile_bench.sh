#!/bin/bash
# rocprofv3 passes over bench.py (run on the GPU box through gpurun):
#   1. --kernel-trace --stats          -> per-kernel durations
#   2. --pmc FETCH_SIZE  (own pass)    -> HBM read side   (TCC slots: FETCH 3 of 4)
#   3. --pmc WRITE_SIZE  (own pass)    -> HBM write side
# Outputs under gpurun_out/prof_$1/ ; summaries are copied to profiles/ by tools/summarize_profile.py
set -e
TAG=${1:-r03}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
rm -rf $OUT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $OUT/bench_trace.json 2> $OUT/trace.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > $OUT/bench_pmc_fetch.json 2> $OUT/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > $OUT/bench_pmc_write.json 2> $OUT/pmc_write.err
find $OUT -name "*.csv" | head -20
du -sh $OUT
