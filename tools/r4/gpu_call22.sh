#!/bin/bash
# round 4, call 22 (no library change behind it): the matrix-core walk's parity cases five times over, looking for a rare race
OUT=gpurun_out/r4c22
mkdir -p $OUT
export TMPDIR=/tmp
for i in 1 2 3 4 5; do
  timeout 300 python -m pytest tests/test_decode_kernels_gpu.py tests/test_zz_attn_qkv_partials_gpu.py tests/test_zz_engine_windows_vs_truth_gpu.py -q -x -p no:cacheprovider -k "long_contexts or matrix_core or gqa_group" 2>&1 | tail -2 | tee -a $OUT/pytest_repeats.txt
done
echo done
