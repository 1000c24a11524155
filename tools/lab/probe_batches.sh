#!/bin/bash
# usage (on the GPU box): bash tools/lab/probe_batches.sh <batch> [...]: rocprofv3 kernel stats of eager batched decode steps
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for b in "$@"; do
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_b$b -o b$b --output-format csv -- python $R/tools/batch_decode_probe.py --batch $b --context 256 --no-graph --steps 8 > $R/gpurun_out/prof_b$b.log 2>&1
done
