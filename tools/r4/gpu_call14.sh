#!/bin/bash
# round 4, call 14: two attention workgroups per CU (64 / 128 windows) at 8k and 32k now that a stage's waits are exact
OUT=gpurun_out/r4c14
mkdir -p $OUT
export TMPDIR=/tmp
timeout 500 python tools/decode_ab.py --batch 1 --prompt-len 8000 --steps 128 - TL_ATTN_MAX_SPLITS=64,TL_ATTN_MIN_TOKENS=128 TL_ATTN_MIN_TOKENS=512 - TL_ATTN_MAX_SPLITS=64,TL_ATTN_MIN_TOKENS=128 2>&1 | grep -v Warning | tee $OUT/ab_8k.jsonl | cut -c1-420
timeout 900 python tools/decode_ab.py --batch 1 --prompt-len 32000 --steps 128 - TL_ATTN_MAX_SPLITS=64 TL_ATTN_MAX_SPLITS=128 - TL_ATTN_MAX_SPLITS=64 2>&1 | grep -v Warning | tee $OUT/ab_32k.jsonl | cut -c1-420
echo done
