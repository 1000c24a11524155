#!/bin/bash
# r02 call G: in-kernel merge of the decode-attention splits -- parity, A/B against the merge launch, bench configs.
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/call_g
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_decode_kernels_gpu.py tests/test_engine_gpu.py tests/test_engine_qwen4b_gpu.py -m gpu -q --tb=short -p no:cacheprovider -x -k "attention or engine" > $OUT/pytest_attn.log 2>&1
echo "pytest attention/engine rc=$?"; tail -5 $OUT/pytest_attn.log | cut -c1-300
for FM in 1 0; do
  TL_ATTN_FUSED_MERGE=$FM timeout 600 python bench.py --no-cpu-baseline > $OUT/bench_fm$FM.json 2> $OUT/bench_fm$FM.err
  for B in 4 16; do
    TL_ATTN_FUSED_MERGE=$FM timeout 600 python tools/decode_ab.py --batch $B --prompt-len 256 --steps 128 --profile-steps 4 - >> $OUT/ab_fm$FM.jsonl 2>> $OUT/ab.err
  done
done
TL_ATTN_FUSED_MERGE=1 timeout 600 python bench.py --config 3 --no-cpu-baseline > $OUT/bench_c3.json 2> $OUT/bench_c3.err
python - <<'PY'
import json
for c in ("bench_fm1","bench_fm0","bench_c3"):
    try:
        b=json.loads(open(f"gpurun_out/call_g/{c}.json").read().strip().splitlines()[-1]); r=b["roofline"]
        print(c,b["value"],b["ms_per_step"],"launches",r["launches_per_step_all_kernels"],{k:v["us_per_step"] for k,v in r["per_kind"].items() if "att" in k})
    except Exception as e: print(c,"failed",e)
for f in ("ab_fm1","ab_fm0"):
    for l in open(f"gpurun_out/call_g/{f}.jsonl"):
        r=json.loads(l); print(f,"batch",r["batch"],"ms/step",r["ms_per_step"],"launches",r.get("launches"),"splits",r.get("n_splits"))
PY
