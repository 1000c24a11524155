#!/bin/bash
# round 4, call 17: split counts for the matrix-core walk (two workgroups per CU), and short windows at few sequences
OUT=gpurun_out/r4c17
mkdir -p $OUT
export TMPDIR=/tmp
timeout 400 python tools/decode_ab.py --batch 1 --prompt-len 32000 --steps 64 - TL_ATTN_MAX_SPLITS=64 TL_ATTN_MAX_SPLITS=128 - TL_ATTN_MAX_SPLITS=64 2>&1 | grep -v Warning | tee -a $OUT/ab.jsonl | cut -c1-300
timeout 300 python tools/decode_ab.py --batch 1 --prompt-len 8000 --steps 64 - TL_ATTN_MAX_SPLITS=64,TL_ATTN_MIN_TOKENS=128 TL_ATTN_MIN_TOKENS=512 - TL_ATTN_MAX_SPLITS=64,TL_ATTN_MIN_TOKENS=128 2>&1 | grep -v Warning | tee -a $OUT/ab.jsonl | cut -c1-300
for b in 4 8; do timeout 300 python tools/decode_ab.py --batch $b --prompt-len 200 --steps 64 - TL_ATTN_MFMA=0 - TL_ATTN_MFMA=0 2>&1 | grep -v Warning | tee -a $OUT/ab.jsonl | cut -c1-300; done
timeout 300 python tools/decode_ab.py --batch 4 --prompt-len 2000 --steps 64 - TL_ATTN_MFMA=0 - TL_ATTN_MFMA=0 2>&1 | grep -v Warning | tee -a $OUT/ab.jsonl | cut -c1-300
echo done
