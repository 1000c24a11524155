#!/bin/bash
OUT=gpurun_out/r4c8
mkdir -p $OUT
export TMPDIR=/tmp
for b in 8 16; do
  timeout 300 python tools/decode_ab.py --batch $b --prompt-len 256 --steps 64 TL_QMM5=0 TL_QMM5=0,TL_QMM3_STAGE_FIRST=0 TL_QMM5=0 TL_QMM5=0,TL_QMM3_STAGE_FIRST=0 2>&1 | grep -v Warning | tee -a $OUT/stage_first_ab2.jsonl
done
timeout 300 python tools/decode_ab.py --batch 32 --prompt-len 256 --steps 64 TL_QMM5=0 TL_QMM5=0,TL_QMM3_STAGE_FIRST=1 TL_QMM5=0 TL_QMM5=0,TL_QMM3_STAGE_FIRST=1 2>&1 | grep -v Warning | tee -a $OUT/stage_first_ab2.jsonl
timeout 300 python tools/decode_ab.py --batch 5 --prompt-len 256 --steps 64 TL_QMM5=0 TL_QMM5=0,TL_QMM3_STAGE_FIRST=0 TL_QMM5=1 2>&1 | grep -v Warning | tee -a $OUT/stage_first_ab2.jsonl
echo done
