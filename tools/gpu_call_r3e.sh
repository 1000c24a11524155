#!/bin/bash
# Round 3, call E: DPP reductions + producer-side sums of squares: full GPU suite, then the single-stream A/B.
OUT=gpurun_out/r3e
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -8 | tee $OUT/tests.log
timeout 300 python tools/decode_ab.py --batch 1 --prompt-len 128 --steps 128 --profile-steps 2 TL_GEMV_PRODUCER_SS=0 - TL_GEMV_PRODUCER_SS=0 - > $OUT/ab.jsonl 2>> $OUT/ab.err
python - <<'PY'
import json
for l in open("gpurun_out/r3e/ab.jsonl"):
    r=json.loads(l); u=r.get("us_per_step",{})
    print(r["batch"],r["variant"],"ms",r["ms_per_step"],"kernel_us",r.get("kernel_us_per_step"),"qkv",u.get("gemv_qkv"),"o",u.get("gemv_o"),"gu",u.get("gemv_gate_up"),"down",u.get("gemv_down"),"lm",u.get("gemv_lm_head"),"attn",u.get("attention"),"merge",u.get("attention_merge"))
PY
