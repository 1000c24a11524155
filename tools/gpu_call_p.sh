#!/bin/bash
# First call of round 3 (everything here was written at the end of round 2 without device time):
#   1. tests/test_zz_attn_qkv_partials_gpu.py (qkv slice partials read by the attention kernel; the three Qwen3-4B-shaped cases
#      have never run) and tests/test_zz_wo_merges_attn_gpu.py (the wo GEMV merging the attention splits; never run);
#   2. the step time with and without TL_ATTN_QKV_PARTIALS=1 at 5 .. 64 sequences (expected: 36 launches and ~0.12 ms fewer),
#      and single stream with and without TL_WO_MERGES_ATTN=1 (expected: 36 launches and ~40-50 us per token fewer);
#   3. decode attention through the MFMA FlashAttention kernel (16-row queries) next to the engine's kernel, 1k .. 32k tokens:
#      what an MFMA decode-attention kernel could sustain (Next, item 3).
# A warm box charges ~12-15 s per gpurun call on top of the command's own time: short calls are cheap.
OUT=gpurun_out/call_p
mkdir -p $OUT
export TMPDIR=/tmp
export TL_UNREHEARSED_GPU_TESTS=1  # the gated tests of the two opt-in routes (a crash here must not take a full pytest run with it)
timeout 300 python -m pytest tests/test_zz_attn_qkv_partials_gpu.py tests/test_zz_wo_merges_attn_gpu.py -q -p no:cacheprovider -rxXfE 2>&1 | tail -20 | tee $OUT/opt_in_route_tests.log
rm -f $OUT/ab.jsonl
run() { B=$1; shift; timeout 300 python tools/decode_ab.py --batch $B --prompt-len 256 --steps 64 --profile-steps 2 "$@" >> $OUT/ab.jsonl 2>> $OUT/ab.err; }
for B in 5 8 16 32 64; do run $B - TL_ATTN_QKV_PARTIALS=1; done
# single stream, bench.py's workload (128-token prompt) and a 4-window context: the merge launch against the merging wo GEMV
run1() { timeout 300 python tools/decode_ab.py --batch 1 --prompt-len $1 --steps 128 --profile-steps 2 - TL_WO_MERGES_ATTN=1 >> $OUT/ab.jsonl 2>> $OUT/ab.err; }
run1 128
run1 40
run1 280
python - <<'PY'
import json
for l in open("gpurun_out/call_p/ab.jsonl"):
    r=json.loads(l); u=r.get("us_per_step",{})
    print(r["batch"],r["variant"],"ms",r["ms_per_step"],"launches",r.get("launches"),"kernel_us",r.get("kernel_us_per_step"),"attn",u.get("attention"),"qkv",u.get("gemv_qkv"))
PY
tail -3 $OUT/ab.err
timeout 300 python benches/bench_week3_attention.py --contexts 1024 8192 32768 --mfma-rows --json-output $OUT/attention_mfma_rows.json 2>&1 | tee $OUT/attention_mfma_rows.txt | tail -12
