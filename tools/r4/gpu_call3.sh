#!/bin/bash
# Round 4, third device call: full device suite on the build with the contiguous-page attention route and the lm_head tile maxima;
# single-stream A/B of both; bench.py with its live rocprofv3 leg and the new CPU legs; long-context A/B.
OUT=gpurun_out/r4c3
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -15 | tee $OUT/tests.log
timeout 300 python tools/decode_ab.py --batch 1 --prompt-len 128 --steps 256 - TL_ATTN_CONTIG=0 TL_LMHEAD_TILE_MAX=0 TL_ATTN_CONTIG=0,TL_LMHEAD_TILE_MAX=0 - 2>&1 | grep -v Warning | tee $OUT/single_stream_ab.jsonl
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $OUT/bench.json 2> $OUT/bench.err; tail -c 6000 $OUT/bench.json; tail -5 $OUT/bench.err
for v in 1 0; do
  TL_ATTN_CONTIG=$v timeout 400 python bench.py --config 3 --no-cpu-baseline --rocprof off > $OUT/bench_c3_contig$v.json 2>> $OUT/bench.err
  TL_ATTN_CONTIG=$v timeout 600 python bench.py --config 5 --no-cpu-baseline --rocprof off > $OUT/bench_c5_contig$v.json 2>> $OUT/bench.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4c3/bench_c*_contig*.json")):
    try:
        b=json.loads(open(f).read().strip().splitlines()[-1]); r=b["roofline"]
        print(f, "ms", b["ms_per_step"], "attn", r.get("attention_kv"))
    except Exception as e:
        print(f, "ERR", e)
PY
cp gpurun_out/parity_numbers.jsonl $OUT/ 2>/dev/null
cp -r gpurun_out/bench_rocprof $OUT/ 2>/dev/null
echo done
