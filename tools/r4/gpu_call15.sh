#!/bin/bash
# round 4, call 15: decode attention on the matrix cores (attn_mfma.h) -- parity first, then the batched step against the VALU walk
OUT=gpurun_out/r4c15
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_decode_kernels_gpu.py -q -x -k "attention" 2>&1 | tail -15 | tee $OUT/pytest.txt
timeout 300 python -m pytest tests/test_zz_attn_qkv_partials_gpu.py -q -x 2>&1 | tail -8 | tee -a $OUT/pytest.txt
for b in 64 16; do timeout 300 python tools/decode_ab.py --batch $b --prompt-len 256 --steps 64 - TL_ATTN_MFMA=0 - TL_ATTN_MFMA=0 2>&1 | grep -v Warning | tee -a $OUT/ab.jsonl | cut -c1-600; done
echo done
