#!/bin/bash
# Runs on the GPU box (via gpurun): bench line, rocprofv3 kernel-trace stats of the same command, and two separate
# PMC passes (FETCH_SIZE / WRITE_SIZE cannot share a pass; --pmc is never combined with extra trace domains).
# usage: bash tools/profile_bench.sh <tag>
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $OUT/bench.json 2> $OUT/bench.err
BENCH_ARGS="--steps 64 --warmup 8 --no-cpu-baseline --profile-steps 0"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench --output-format csv -- python $R/bench.py $BENCH_ARGS > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o bench --output-format csv -- python $R/bench.py $BENCH_ARGS > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o bench --output-format csv -- python $R/bench.py $BENCH_ARGS > $OUT/pmc_write.log 2>&1
ls -R $OUT | head -30
cat $OUT/bench.json
