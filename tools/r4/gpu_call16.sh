#!/bin/bash
# round 4, call 16: matrix-core attention walk against the VALU walk where the walk dominates (long contexts, many sequences)
OUT=gpurun_out/r4c16
mkdir -p $OUT
export TMPDIR=/tmp
timeout 400 python tools/decode_ab.py --batch 1 --prompt-len 32000 --steps 64 - TL_ATTN_MFMA=0 - TL_ATTN_MFMA=0 2>&1 | grep -v Warning | tee -a $OUT/ab.jsonl | cut -c1-500
timeout 300 python tools/decode_ab.py --batch 1 --prompt-len 8000 --steps 64 - TL_ATTN_MFMA=0 - TL_ATTN_MFMA=0 2>&1 | grep -v Warning | tee -a $OUT/ab.jsonl | cut -c1-500
timeout 300 python tools/decode_ab.py --batch 16 --prompt-len 2000 --steps 64 - TL_ATTN_MFMA=0 - TL_ATTN_MFMA=0 2>&1 | grep -v Warning | tee -a $OUT/ab.jsonl | cut -c1-500
timeout 400 python tools/decode_ab.py --batch 64 --prompt-len 1000 --steps 64 - TL_ATTN_MFMA=0 - TL_ATTN_MFMA=0 2>&1 | grep -v Warning | tee -a $OUT/ab.jsonl | cut -c1-500
echo done
