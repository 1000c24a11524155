#!/bin/bash
# usage (on the GPU box): bash tools/lab/probe_prefill.sh <chunk> [...]: rocprofv3 kernel stats of an 8k chunked prefill
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in "$@"; do
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_pf$c -o pf$c --output-format csv -- python $R/tools/prefill_probe.py --chunk $c --repeat 0 > $R/gpurun_out/prof_pf$c.log 2>&1
done
