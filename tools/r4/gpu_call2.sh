#!/bin/bash
# Round 4, second device call: in-launch slice reduction with whole-line partial blocks per tile -- parity (bit identity with the
# reduction launch) and A/B at 8 / 16 / 64 rows; the new parity tests in full.
OUT=gpurun_out/r4c2
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_decode_kernels_gpu.py tests/test_zz_engine_windows_vs_truth_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -40 | tee $OUT/tests_new.log
timeout 600 python -m pytest tests/test_engine_qwen4b_gpu.py tests/test_engine_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -8 | tee $OUT/tests_engine.log
for b in 8 16 64; do
  timeout 300 python tools/decode_ab.py --batch $b --prompt-len 256 --steps 64 - TL_QMM3_FIXUP=0 2>&1 | grep -v Warning | tee -a $OUT/fixup_ab.jsonl
done
cp gpurun_out/parity_numbers.jsonl $OUT/ 2>/dev/null
echo done
