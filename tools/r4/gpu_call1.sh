#!/bin/bash
# Round 4, first device call: the new parity tests (headline kernels through tl_decode_linear_ex, engine at 2-16 attention windows
# against the truth fixture, in-launch slice reduction of the skinny matmul) and the A/B of the in-launch reduction at 5-64 rows.
OUT=gpurun_out/r4c1
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_decode_kernels_gpu.py tests/test_zz_engine_windows_vs_truth_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -25 | tee $OUT/tests_new.log
timeout 600 python -m pytest tests/test_engine_qwen4b_gpu.py tests/test_engine_gpu.py tests/test_zz_wo_merges_attn_gpu.py tests/test_zz_gemv_weighted_rows_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -15 | tee $OUT/tests_engine.log
for b in 5 8 16 32 64; do
  timeout 300 python tools/decode_ab.py --batch $b --prompt-len 256 --steps 64 - TL_QMM3_FIXUP=0 - TL_QMM3_FIXUP=0 2>&1 | grep -v Warning | tee -a $OUT/fixup_ab.jsonl
done
timeout 300 python tools/decode_ab.py --batch 1 --prompt-len 128 --steps 256 - - 2>&1 | tee -a $OUT/single_stream.jsonl
cp gpurun_out/parity_numbers.jsonl $OUT/ 2>/dev/null
echo done
