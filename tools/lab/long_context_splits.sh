#!/bin/bash
# usage (on the GPU box): bash tools/lab/long_context_splits.sh "<ENV=.. ENV=..>" ...   decode throughput at a 32k context per setting
for cfg in "$@"; do
  echo "$cfg"
  env $cfg python -m benches.bench --num-seqs 1 --min-input-len 32768 --max-input-len 32768 --min-output-len 33 --max-output-len 33 --prefill-step 2048 --warmup 0 2>&1 | grep -E "Decode throughput"
done
