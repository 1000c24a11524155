#!/bin/bash
# r02 call I: prefill GEMM with fused epilogues + scale prefetch, FA page-id fix -- parity and prefill throughput A/B.
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/call_i
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_ops_gpu.py tests/test_models_gpu.py -m gpu -q --tb=short -p no:cacheprovider -x > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; tail -6 $OUT/pytest.log | cut -c1-300
for FE in 1 0; do
  TL_GEMM_FUSED_EPILOGUE=$FE timeout 600 python bench.py --config 3 --no-cpu-baseline --steps 16 --warmup 4 > $OUT/bench_c3_fe$FE.json 2> $OUT/bench_c3_fe$FE.err
done
python - <<'PY'
import json
for c in ("bench_c3_fe1","bench_c3_fe0"):
    try:
        b=json.loads(open(f"gpurun_out/call_i/{c}.json").read().strip().splitlines()[-1])
        print(c,b["value"],b["ms_per_step"],"prefill tok/s",b["prefill_tokens_per_s"])
    except Exception as e: print(c,"failed",e)
PY
timeout 900 python benches/serve_replicas.py --num-seqs 128 --batch-size 64 --json-output $OUT/replicas.json > $OUT/replicas.log 2>&1
echo "replicas rc=$?"; grep -E "^Time|^Total|^Prefill|^Decode throughput|Decode step p50|Peak active" $OUT/replicas.log
