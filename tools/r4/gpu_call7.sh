#!/bin/bash
# Round 4: A/B of "staging loads before the weights" in the skinny matmul kernels (TL_QMM3_STAGE_FIRST), parity of the variant.
OUT=gpurun_out/r4c7
mkdir -p $OUT
export TMPDIR=/tmp
for b in 8 16 64; do
  timeout 300 python tools/decode_ab.py --batch $b --prompt-len 256 --steps 64 TL_QMM5=0 TL_QMM5=0,TL_QMM3_STAGE_FIRST=1 TL_QMM5=0 TL_QMM5=0,TL_QMM3_STAGE_FIRST=1 2>&1 | grep -v Warning | tee -a $OUT/stage_first_ab.jsonl
done
echo done
