#!/bin/bash
# routing change at 5..8 rows + query-head rule at 3..4 sequences: tests, then the step time at 3..8 sequences
OUT=gpurun_out/call_o
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_decode_kernels_gpu.py tests/test_engine_qwen4b_gpu.py tests/test_engine_gpu.py -m gpu -x -q 2>&1 | tail -4
rm -f $OUT/ab2.jsonl
run() { B=$1; shift; timeout 300 python tools/decode_ab.py --batch $B --prompt-len 256 --steps 64 --profile-steps 2 "$@" >> $OUT/ab2.jsonl 2>> $OUT/ab2.err; }
run 3 - TL_ATTN_RQ1_BATCH=4 TL_QMM3_MIN_M=3
run 4 - TL_QMM3_MIN_M=3
run 5 -
run 6 -
run 8 -
python - <<'PY'
import json
for l in open("gpurun_out/call_o/ab2.jsonl"):
    r=json.loads(l); u=r.get("us_per_step",{})
    print(r["batch"],r["variant"],"ms",r["ms_per_step"],"splits",r.get("n_splits"),"launches",r.get("launches"),"attn",u.get("attention"),"qkv",u.get("gemv_qkv"),"o",u.get("gemv_o"),"gu",u.get("gemv_gate_up"),"down",u.get("gemv_down"))
PY
tail -3 $OUT/ab2.err
