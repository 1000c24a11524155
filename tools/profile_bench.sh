#!/bin/bash
# Runs on the GPU box (via gpurun): bench line and rocprofv3 kernel-trace stats of the same command (eager launches).
# usage: bash tools/profile_bench.sh <tag>
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $OUT/bench.json 2> $OUT/bench.err
BENCH_ARGS="--steps 64 --warmup 8 --no-cpu-baseline --profile-steps 0"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench --output-format csv -- python $R/bench.py $BENCH_ARGS > $OUT/trace.log 2>&1
# PMC passes (FETCH_SIZE / WRITE_SIZE, one counter per pass) run over the standalone GEMV lab: tools/lab/pmc_gemv.sh
ls -R $OUT | head -30
cat $OUT/bench.json
