#!/bin/bash
# round 4, call 12: per-projection routing of the batched step (qkv / wo by row count) -- engine parity at batch sizes + same-box A/B
OUT=gpurun_out/r4c12
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_zz_attn_qkv_partials_gpu.py tests/test_engine_qwen4b_gpu.py tests/test_engine_gpu.py tests/test_zz_batched_matmul_gpu.py -q 2>&1 | tail -8 | tee $OUT/pytest.txt
for b in 32 64 24 16 8; do
  timeout 300 python tools/decode_ab.py --batch $b --prompt-len 128 --steps 64 - TL_NO_QMM6=1 - TL_NO_QMM6=1 2>&1 | grep -v Warning | tee -a $OUT/qmm6_ab.jsonl
done
echo done
