#!/bin/bash
# On the GPU box: the engine's decode step at long contexts over bf16 / FP8 pages, default attention plan against 64 windows
# (tools/lab/engine_step_lab: C ABI only).  usage: tools/r6_kv8_lab.sh
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/kv8_lab.log; : > $OUT
for ctx in 8192 32768; do
  for kv in 0 1; do
    for sp in default 64; do
      if [ $sp = default ]; then unset TL_ATTN_MAX_SPLITS; else export TL_ATTN_MAX_SPLITS=$sp; fi
      echo "== ctx $ctx kv $kv splits $sp" >> $OUT
      timeout 180 $R/tools/lab/engine_step_lab $ctx 100 1 $kv 2>&1 | grep "^engine" >> $OUT
    done
  done
done
unset TL_ATTN_MAX_SPLITS
cat $OUT
