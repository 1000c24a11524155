#!/bin/bash
# Round 4, sixth device call: the full-row persistent matmul (qmm5) -- parity at the real shapes, engine tests, A/B at 5 / 8 / 16 rows.
OUT=gpurun_out/r4c6
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_decode_kernels_gpu.py -m gpu -q -p no:cacheprovider -k "full_row or routing or skinny" 2>&1 | tail -25 | tee $OUT/tests_qmm5.log
timeout 600 python -m pytest tests/test_engine_qwen4b_gpu.py tests/test_engine_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -8 | tee $OUT/tests_engine.log
for b in 5 8 16; do
  timeout 300 python tools/decode_ab.py --batch $b --prompt-len 256 --steps 64 - TL_QMM5=0 TL_QMM5=2 - TL_QMM5=0 2>&1 | grep -v Warning | tee -a $OUT/qmm5_ab.jsonl
done
cp gpurun_out/parity_numbers.jsonl $OUT/ 2>/dev/null
echo done
