#!/bin/bash
# Round 4, fifth device call: bench.py with hard time boxes around every CPU leg and timestamped progress on stderr.
OUT=gpurun_out/r4c5
mkdir -p $OUT
export TMPDIR=/tmp
nproc > $OUT/host.txt; lscpu | head -20 >> $OUT/host.txt; free -g | head -2 >> $OUT/host.txt
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $OUT/bench.json 2> $OUT/bench.err; tail -c 9000 $OUT/bench.json; cat $OUT/bench.err | tail -30
cp -r gpurun_out/bench_rocprof $OUT/ 2>/dev/null
echo done
