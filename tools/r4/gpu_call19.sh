#!/bin/bash
# round 4, call 19: K/V requests of the matrix-core walk as buffer loads (exact waits in both stages) -- parity, then 32k / 8k / batched
OUT=gpurun_out/r4c19
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_decode_kernels_gpu.py -q -x -k "long_contexts and default" 2>&1 | tail -4 | tee $OUT/pytest.txt
timeout 600 python -m pytest tests/test_zz_attn_qkv_partials_gpu.py tests/test_zz_engine_windows_vs_truth_gpu.py -q -x 2>&1 | tail -4 | tee -a $OUT/pytest.txt
timeout 400 python tools/decode_ab.py --batch 1 --prompt-len 32000 --steps 64 - TL_ATTN_MAX_SPLITS=32 - TL_ATTN_MAX_SPLITS=32 2>&1 | grep -v Warning | tee -a $OUT/ab.jsonl | cut -c1-330
timeout 300 python tools/decode_ab.py --batch 1 --prompt-len 8000 --steps 64 - - 2>&1 | grep -v Warning | tee -a $OUT/ab.jsonl | cut -c1-330
timeout 300 python tools/decode_ab.py --batch 16 --prompt-len 2000 --steps 64 - - 2>&1 | grep -v Warning | tee -a $OUT/ab.jsonl | cut -c1-330
timeout 400 python tools/decode_ab.py --batch 64 --prompt-len 1000 --steps 64 - - 2>&1 | grep -v Warning | tee -a $OUT/ab.jsonl | cut -c1-330
echo done
