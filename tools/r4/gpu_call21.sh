#!/bin/bash
# round 4, call 21 (no library change behind it): the driver's own launch shapes on the final build
OUT=gpurun_out/r4c21
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_torchrun_n1.json 2> $OUT/bench_torchrun_n1.err; echo "torchrun rc=$?"; tail -1 $OUT/bench_torchrun_n1.json | cut -c1-400
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --rocprof off > $OUT/bench_plain_n1.json 2> $OUT/bench_plain_n1.err; echo "plain rc=$?"; tail -1 $OUT/bench_plain_n1.json | cut -c1-200
echo done
