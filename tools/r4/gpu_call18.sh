#!/bin/bash
# round 4, call 18: the new twin tests of the matrix-core walk; where one query head per workgroup should hand over to the GQA-group walk
OUT=gpurun_out/r4c18
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_decode_kernels_gpu.py -q -x -k "long_contexts" 2>&1 | tail -6 | tee $OUT/pytest.txt
timeout 600 python -m pytest tests/test_zz_attn_qkv_partials_gpu.py tests/test_zz_engine_windows_vs_truth_gpu.py -q -x 2>&1 | tail -8 | tee -a $OUT/pytest.txt
for pl in 700 1500 3000; do timeout 300 python tools/decode_ab.py --batch 1 --prompt-len $pl --steps 64 - TL_ATTN_RQ=4 - TL_ATTN_RQ=4 2>&1 | grep -v Warning | tee -a $OUT/ab.jsonl | cut -c1-330; done
for pl in 700 1500; do timeout 300 python tools/decode_ab.py --batch 2 --prompt-len $pl --steps 64 - TL_ATTN_RQ=4 - TL_ATTN_RQ=4 2>&1 | grep -v Warning | tee -a $OUT/ab.jsonl | cut -c1-330; done
echo done
