#!/bin/bash
# round 4, call 11: whole GPU suite on the build with the register-resident batched matmul + same-box A/B per step
OUT=gpurun_out/r4c11
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $OUT/pytest_gpu.txt
for b in 64 32 16 8 5; do
  timeout 300 python tools/decode_ab.py --batch $b --prompt-len 128 --steps 64 - TL_NO_QMM6=1 - TL_NO_QMM6=1 2>&1 | grep -v Warning | tee -a $OUT/qmm6_ab.jsonl
done
echo done
