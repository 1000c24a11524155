#!/bin/bash
# Round 3, call D: producer-side sums of squares for the 1-4-row GEMVs: engine parity tests, then the single-stream A/B.
OUT=gpurun_out/r3d
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_engine_gpu.py tests/test_engine_qwen4b_gpu.py tests/test_decode_kernels_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -8 | tee $OUT/tests.log
rm -f $OUT/ab.jsonl
timeout 300 python tools/decode_ab.py --batch 1 --prompt-len 128 --steps 128 --profile-steps 2 TL_GEMV_PRODUCER_SS=0 - TL_GEMV_PRODUCER_SS=0 - >> $OUT/ab.jsonl 2>> $OUT/ab.err
timeout 300 python tools/decode_ab.py --batch 2 --prompt-len 128 --steps 64 --profile-steps 2 TL_GEMV_PRODUCER_SS=0 - >> $OUT/ab.jsonl 2>> $OUT/ab.err
timeout 300 python tools/decode_ab.py --batch 4 --prompt-len 128 --steps 64 --profile-steps 2 TL_GEMV_PRODUCER_SS=0 - >> $OUT/ab.jsonl 2>> $OUT/ab.err
python - <<'PY'
import json
for l in open("gpurun_out/r3d/ab.jsonl"):
    r=json.loads(l); u=r.get("us_per_step",{})
    print(r["batch"],r["variant"],"ms",r["ms_per_step"],"launches",r.get("launches"),"kernel_us",r.get("kernel_us_per_step"),"qkv",u.get("gemv_qkv"),"o",u.get("gemv_o"),"gu",u.get("gemv_gate_up"),"down",u.get("gemv_down"),"lm",u.get("gemv_lm_head"))
PY
tail -3 $OUT/ab.err
