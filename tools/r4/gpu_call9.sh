#!/bin/bash
# round 4, call 9: first run of the register-resident batched matmul (csrc/qmm6.h): kernel parity, engine parity, A/B per step
OUT=gpurun_out/r4c9
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_zz_batched_matmul_gpu.py -x -q 2>&1 | tail -25 | tee $OUT/pytest_qmm6.txt
timeout 600 python -m pytest tests/test_engine_qwen4b_gpu.py -q -k "batched or tile_maxima" 2>&1 | tail -15 | tee $OUT/pytest_engine_batched.txt
for b in 8 64 16 32 5; do
  timeout 300 python tools/decode_ab.py --batch $b --prompt-len 128 --steps 64 - TL_NO_QMM6=1 - TL_NO_QMM6=1 2>&1 | grep -v Warning | tee -a $OUT/qmm6_ab.jsonl
done
echo done
