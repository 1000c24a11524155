#!/bin/bash
# On the GPU box: rocprofv3 kernel trace of a chunked prefill (8k prompt, 4,096-token chunks) -> gpurun_out/prefill_trace/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prefill_trace -o pf --output-format csv -- python $R/tools/prefill_probe.py --prompt 8192 --chunk 4096 --repeat 1 > $R/gpurun_out/prefill_trace.log 2>&1
tail -2 $R/gpurun_out/prefill_trace.log
