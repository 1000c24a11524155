#!/bin/bash
# Evidence run of a round (copied into profiles/ by tools/collect_evidence.py): whole GPU suite (with the reference's own tests when staged:
# tools/stage_reference_tests.sh), smoke, bench configs 2/3/5 (+ rocprofv3 kernel stats of each), batched decode
# table, serving (one GPU, reference admission and packed admission), acceptance run, operator / attention microbenches.
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/evidence
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
rm -f $R/gpurun_out/parity_numbers.jsonl
timeout 1800 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --durations=8 > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
grep -E "^E   |^FAILED|passed|failed" $OUT/pytest.log | cut -c1-300 | tail -20
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
timeout 600 python bench.py --config 3 --no-cpu-baseline > $OUT/bench_c3.json 2> $OUT/bench_c3.err
timeout 600 python bench.py --config 5 --no-cpu-baseline > $OUT/bench_c5.json 2> $OUT/bench_c5.err
python - <<'PY'
import json
for c in ("bench","bench_c3","bench_c5"):
    try:
        b=json.loads(open(f"gpurun_out/evidence/{c}.json").read().strip().splitlines()[-1]); r=b["roofline"]
        print(c,b["value"],b["ms_per_step"],"prefill",b["prefill_tokens_per_s"],"frac",r["frac"],"step_frac",r["step_frac"],r.get("rocprof",{}).get("frac"),r["attention_kv"]["frac"])
    except Exception as e: print(c,"failed",e)
PY
rm -f $OUT/ab_batched.jsonl
for B in 2 4 8 16 32 64; do
  timeout 600 python tools/decode_ab.py --batch $B --prompt-len 256 --steps 128 --profile-steps 4 - >> $OUT/ab_batched.jsonl 2>> $OUT/ab_batched.err
done
python - <<'PY'
import json
for l in open("gpurun_out/evidence/ab_batched.jsonl"):
    r=json.loads(l); print("batch",r["batch"],"ms/step",r["ms_per_step"],"tok/s",r["tokens_per_s"],"launches",r.get("launches"))
PY
timeout 900 python benches/serve_replicas.py --num-seqs 128 --batch-size 64 --json-output $OUT/replicas_n1.json > $OUT/replicas_n1.log 2>&1
echo "replicas rc=$?"; grep -E "^Time|^Total|^Prefill|^Decode throughput|Decode step p50|Peak active" $OUT/replicas_n1.log
timeout 900 python benches/serve_replicas.py --num-seqs 128 --batch-size 64 --staging-slots 1 --prefill-step 128 --json-output $OUT/replicas_n1_reference_admission.json > $OUT/replicas_n1_reference_admission.log 2>&1
grep -E "^Time|^Total|^Prefill|^Decode throughput" $OUT/replicas_n1_reference_admission.log
timeout 600 python -m benches.bench --num-seqs 1 --min-input-len 128 --max-input-len 128 --min-output-len 129 --max-output-len 129 --prefill-logits last --warmup 2 --json-output $OUT/acceptance.json > $OUT/acceptance.log 2>&1; tail -5 $OUT/acceptance.log
timeout 600 python benches/bench_week2_operators.py --json-output $OUT/operators.json > $OUT/operators.log 2>&1; tail -12 $OUT/operators.log
timeout 600 python benches/bench_week3_attention.py --json-output $OUT/attention.json > $OUT/attention.log 2>&1; tail -6 $OUT/attention.log
cd /tmp
for C in 2 3 5; do
  rocprofv3 --kernel-trace --stats -d $OUT/trace_c$C -o bench --output-format csv -- python $R/bench.py --config $C --steps 16 --warmup 4 --no-cpu-baseline --profile-steps 0 > $OUT/trace_c$C.log 2>&1
  echo "trace c$C rc=$?"
done
rocprofv3 --kernel-trace --stats -d $OUT/trace_b64 -o b64 --output-format csv -- python $R/tools/decode_ab.py --batch 64 --prompt-len 256 --steps 32 --profile-steps 0 - > $OUT/trace_b64.log 2>&1
echo "trace b64 rc=$?"
cp $R/gpurun_out/parity_numbers.jsonl $OUT/ 2>/dev/null
