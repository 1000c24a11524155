#!/bin/bash
# Round 4, fourth device call: the full device suite on the pruned build and the driver's bench command with its bounded CPU legs.
OUT=gpurun_out/r4c4
mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -25 ) 2>&1 | tee $OUT/tests.log
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $OUT/bench.json 2> $OUT/bench.err; tail -c 7000 $OUT/bench.json; tail -4 $OUT/bench.err
cp gpurun_out/parity_numbers.jsonl $OUT/ 2>/dev/null
cp -r gpurun_out/bench_rocprof $OUT/ 2>/dev/null
echo done
