#!/bin/bash
# round 4, call 20: 128-token windows (matrix-core walk) against 64-token windows (VALU walk) for 3-11 sequences at short contexts
OUT=gpurun_out/r4c20
mkdir -p $OUT
export TMPDIR=/tmp
for b in 4 8; do for pl in 180 400; do timeout 300 python tools/decode_ab.py --batch $b --prompt-len $pl --steps 64 - TL_ATTN_MIN_TOKENS=128 - TL_ATTN_MIN_TOKENS=128 2>&1 | grep -v Warning | tee -a $OUT/ab.jsonl | cut -c1-200; done; done
echo done
