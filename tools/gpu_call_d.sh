#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/call_d
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_decode_kernels_gpu.py tests/test_engine_gpu.py tests/test_engine_qwen4b_gpu.py -m gpu -q --tb=short -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
grep -E "^E   |^FAILED|passed|failed" $OUT/pytest.log | cut -c1-300 | tail -30
rm -f $OUT/ab_batched.jsonl
for B in 8 16 32 64; do
  timeout 600 python tools/decode_ab.py --batch $B --prompt-len 256 --steps 64 --profile-steps 2 - TL_QMM3_FUSED_NORM=0 >> $OUT/ab_batched.jsonl 2>> $OUT/ab_batched.err
done
python - <<'PY'
import json
for l in open("gpurun_out/call_d/ab_batched.jsonl"):
    r=json.loads(l); print(r["variant"],"batch",r["batch"],"ms/step",r["ms_per_step"],"tok/s",r["tokens_per_s"],r.get("us_per_step"))
PY
timeout 600 python bench.py --config 3 --no-cpu-baseline > $OUT/bench_c3.json 2> $OUT/bench_c3.err
timeout 600 python bench.py --config 5 --no-cpu-baseline > $OUT/bench_c5.json 2> $OUT/bench_c5.err
TL_ATTN_MAX_SPLITS=128 timeout 600 python bench.py --config 5 --no-cpu-baseline > $OUT/bench_c5_s128.json 2> $OUT/bench_c5_s128.err
python - <<'PY'
import json
for c in ("bench_c3","bench_c5","bench_c5_s128"):
    try:
        b=json.loads(open(f"gpurun_out/call_d/{c}.json").read().strip().splitlines()[-1]); r=b["roofline"]
        print(c,b["value"],b["ms_per_step"],"prefill",b["prefill_tokens_per_s"],r["attention_kv"],{k:v["us_per_step"] for k,v in r["per_kind"].items()})
    except Exception as e: print(c,"failed",e)
PY
timeout 900 python -m benches.bench --batch-decode --batch-size 64 --num-seqs 128 --min-input-len 128 --max-input-len 1024 --min-output-len 32 --max-output-len 128 --prefill-step 128 --prefill-budget 2048 --json-output $OUT/serving_b64.json > $OUT/serving_b64.log 2>&1
echo "serving rc=$?"; head -8 $OUT/serving_b64.log; grep -E "Peak active|Decode step latency" $OUT/serving_b64.log
