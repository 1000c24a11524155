#!/bin/bash
# Round 4 evidence run on the final build (copied into profiles/ by tools/r4/collect.py): whole GPU suite, smoke, bench configs 2 / 3 / 5
# (each with its own live rocprofv3 child: gpurun_out/bench_rocprof/), the driver's command shape, the batched decode table and its
# rocprofv3 summary at 64 sequences, serving on one GPU with the driver-shaped JSON line, operator / attention microbenches.
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/evidence_r4
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
rm -f $R/gpurun_out/parity_numbers.jsonl; rm -rf $R/gpurun_out/bench_rocprof
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --durations=8 > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
grep -E "^E   |^FAILED|passed|failed" $OUT/pytest.log | cut -c1-300 | tail -20
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; grep "bench +" $OUT/bench.err | tail -12
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_driver_shape.json 2> $OUT/bench_driver_shape.err
timeout 600 python bench.py --config 3 --no-cpu-baseline > $OUT/bench_c3.json 2> $OUT/bench_c3.err
timeout 900 python bench.py --config 5 --no-cpu-baseline > $OUT/bench_c5.json 2> $OUT/bench_c5.err
python - <<'PY'
import json
for c in ("bench","bench_driver_shape","bench_c3","bench_c5"):
    try:
        b=json.loads(open(f"gpurun_out/evidence_r4/{c}.json").read().strip().splitlines()[-1]); r=b["roofline"]
        print(c,b["value"],b["ms_per_step"],"prefill",b["prefill_tokens_per_s"],"frac",r["frac"],r.get("frac_source","")[:40],"stamps",r.get("frac_in_kernel_stamps"),"step_frac",r["step_frac"],"kv",r["attention_kv"]["frac"])
        if b.get("cpu_baseline"):
            c2=b["cpu_baseline"]; print("  cpu", c2.get("value"), c2.get("sample"), c2.get("n_splits_checked"), c2.get("n_splits_timed"), "torch", c2.get("torch_week2_kv_cache"), "peaked", {k:v for k,v in (c2.get("peaked_checkpoint") or {}).items() if k not in ("recipe",)})
    except Exception as e: print(c,"failed",e)
PY
rm -f $OUT/ab_batched.jsonl
for B in 2 4 8 16 32 64; do
  timeout 600 python tools/decode_ab.py --batch $B --prompt-len 256 --steps 128 --profile-steps 4 - >> $OUT/ab_batched.jsonl 2>> $OUT/ab_batched.err
done
python - <<'PY'
import json
for l in open("gpurun_out/evidence_r4/ab_batched.jsonl"):
    r=json.loads(l); print("batch",r["batch"],"ms/step",r["ms_per_step"],"tok/s",r["tokens_per_s"],"launches",r.get("launches"))
PY
rm -f $OUT/ab_qmm6.jsonl
for B in 8 16 32 64; do
  timeout 300 python tools/decode_ab.py --batch $B --prompt-len 128 --steps 64 - TL_NO_QMM6=1 - TL_NO_QMM6=1 >> $OUT/ab_qmm6.jsonl 2>> $OUT/ab_qmm6.err
done
python - <<'PY'
import json
for l in open("gpurun_out/evidence_r4/ab_qmm6.jsonl"):
    r=json.loads(l); print("batch",r["batch"],r["variant"],"ms/step",r["ms_per_step"],"launches",r.get("launches"))
PY
rm -f $OUT/ab_attn_mfma.jsonl
timeout 400 python tools/decode_ab.py --batch 1 --prompt-len 32000 --steps 64 - TL_ATTN_MFMA=0 - TL_ATTN_MFMA=0 >> $OUT/ab_attn_mfma.jsonl 2>> $OUT/ab_attn_mfma.err
timeout 300 python tools/decode_ab.py --batch 1 --prompt-len 8000 --steps 64 - TL_ATTN_MFMA=0 - TL_ATTN_MFMA=0 >> $OUT/ab_attn_mfma.jsonl 2>> $OUT/ab_attn_mfma.err
timeout 300 python tools/decode_ab.py --batch 16 --prompt-len 2000 --steps 64 - TL_ATTN_MFMA=0 - TL_ATTN_MFMA=0 >> $OUT/ab_attn_mfma.jsonl 2>> $OUT/ab_attn_mfma.err
timeout 400 python tools/decode_ab.py --batch 64 --prompt-len 1000 --steps 64 - TL_ATTN_MFMA=0 - TL_ATTN_MFMA=0 >> $OUT/ab_attn_mfma.jsonl 2>> $OUT/ab_attn_mfma.err
python - <<'PY'
import json
for l in open("gpurun_out/evidence_r4/ab_attn_mfma.jsonl"):
    r=json.loads(l); print("batch",r["batch"],"prompt",r["prompt"],r["variant"],"ms/step",r["ms_per_step"],"attention",r["us_per_step"]["attention"],"merge",r["us_per_step"]["attention_merge"])
PY
if [ -x tools/lab/qmm6_lab_abl0 ]; then for m in 64 32 16 8; do tools/lab/qmm6_lab_abl0 $m 1; tools/lab/qmm6_lab_abl0 $m 0; done > $OUT/qmm6_lab.txt 2>&1; fi
timeout 900 python benches/serve_replicas.py --num-seqs 128 --batch-size 64 --json-output $OUT/replicas_n1.json > $OUT/replicas_n1.log 2>&1
echo "replicas rc=$?"; grep -E "^Time|^Total|^Prefill|^Decode throughput|Decode step p50" $OUT/replicas_n1.log; tail -1 $OUT/replicas_n1.log | cut -c1-400
timeout 600 python benches/bench_week2_operators.py --json-output $OUT/operators.json > $OUT/operators.log 2>&1; tail -12 $OUT/operators.log
timeout 600 python benches/bench_week3_attention.py --json-output $OUT/attention.json > $OUT/attention.log 2>&1; tail -6 $OUT/attention.log
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace_b64 -o b64 --output-format csv -- python $R/tools/decode_ab.py --batch 64 --prompt-len 256 --steps 32 --profile-steps 0 - > $OUT/trace_b64.log 2>&1
echo "trace b64 rc=$?"
rocprofv3 --kernel-trace --stats -d $OUT/trace_b8 -o b8 --output-format csv -- python $R/tools/decode_ab.py --batch 8 --prompt-len 256 --steps 32 --profile-steps 0 - > $OUT/trace_b8.log 2>&1
echo "trace b8 rc=$?"
# HBM bytes per GEMV launch: PMC passes of their own (counters only, no trace domain), tools/make_traffic_json.py reads them
rm -rf $R/gpurun_out/pmc_gemv
if [ -x $R/tools/lab/gemv_lab ]; then timeout 600 bash $R/tools/lab/pmc_gemv.sh > $OUT/pmc_gemv.log 2>&1; echo "pmc rc=$?"; fi
find $R/gpurun_out/pmc_gemv -type f ! -name "lab_counter_collection.csv" ! -name "*.log" -delete 2>/dev/null
du -sh $R/gpurun_out/pmc_gemv 2>/dev/null
cd $R
cp $R/gpurun_out/parity_numbers.jsonl $OUT/ 2>/dev/null
cp -r $R/gpurun_out/bench_rocprof $OUT/ 2>/dev/null
find $OUT -name "*.csv" | grep -v kernel_stats | xargs rm -f 2>/dev/null
find $OUT -name "*kernel_trace*" | xargs rm -f 2>/dev/null
du -sh $OUT
