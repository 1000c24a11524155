#!/bin/bash
# On the GPU box: batched decode steps under a cap on the attention's context splits (TL_ATTN_MAX_SPLITS; 0 = the planner's own choice), by sequence count and context.
# usage: tools/lab/ab_attn_splits_batched.sh "<batches>" "<contexts>" "<caps>"
for ctx in ${2:-128}; do for b in ${1:-8}; do for ms in ${3:-0 1 2}; do
  if [ $ms = 0 ]; then unset TL_ATTN_MAX_SPLITS; else export TL_ATTN_MAX_SPLITS=$ms; fi
  python tools/batch_decode_probe.py --batch $b --context $ctx --steps 48 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(json.dumps({'context':$ctx,'batch':$b,'max_splits':$ms,'ms_per_step':d['ms_per_step']}))"
done; done; done
