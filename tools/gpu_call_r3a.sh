#!/bin/bash
# Round 3, call A: the run-ahead prefetcher lab (tools/lab/prefetch_lab.hip) and the two routes written at the end of round 2
# without device time: their gated tests, then the single-stream A/B of TL_WO_MERGES_ATTN.
OUT=gpurun_out/r3a
mkdir -p $OUT
export TMPDIR=/tmp
timeout 120 tools/lab/prefetch_lab 2>&1 | tee $OUT/prefetch_lab.log
export TL_UNREHEARSED_GPU_TESTS=1
timeout 300 python -m pytest tests/test_zz_wo_merges_attn_gpu.py tests/test_zz_attn_qkv_partials_gpu.py -q -p no:cacheprovider -rxXfE -x 2>&1 | tail -15 | tee $OUT/opt_in_route_tests.log
rm -f $OUT/ab.jsonl
for PL in 128 40 280; do
  timeout 300 python tools/decode_ab.py --batch 1 --prompt-len $PL --steps 128 --profile-steps 2 - TL_WO_MERGES_ATTN=1 >> $OUT/ab.jsonl 2>> $OUT/ab.err
done
python - <<'PY'
import json
for l in open("gpurun_out/r3a/ab.jsonl"):
    r=json.loads(l); u=r.get("us_per_step",{})
    print(r["batch"],r.get("prompt_len"),r["variant"],"ms",r["ms_per_step"],"launches",r.get("launches"),"kernel_us",r.get("kernel_us_per_step"),"attn",u.get("attention"),"merge",u.get("attention_merge"),"o",u.get("gemv_o"))
PY
tail -3 $OUT/ab.err
