#!/bin/bash
# usage (on the GPU box): bash tools/lab/run_attn_variants.sh "<ENV=..> <ENV=..>" ...   one bench run per argument
for cfg in "$@"; do
  tag=$(echo "$cfg" | tr ' =' '__')
  env $cfg python bench.py --no-cpu-baseline > gpurun_out/bench_$tag.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/bench_$tag.json')); pk=d['roofline']['per_kind']
print('$cfg:', 'tok/s', d['value'], 'ms/step', d['ms_per_step'], 'attn us', pk['attention']['us_per_step'], 'merge us', pk['attention_merge']['us_per_step'])"
done
