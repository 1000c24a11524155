#!/bin/bash
# r02 call F: persistent skinny matmul -- parity tests of both grids, the whole GPU suite, batched decode table with the
# one-shot grid (TL_QMM3_PERSISTENT=0) beside the default, serving loop at 64 slots, rocprofv3 trace of 64 sequences.
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/call_f
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
rm -f $R/gpurun_out/parity_numbers.jsonl
timeout 600 python -m pytest tests/test_decode_kernels_gpu.py -m gpu -q --tb=short -p no:cacheprovider -x -k "skinny or routing" > $OUT/pytest_skinny.log 2>&1
echo "pytest skinny rc=$?"; tail -3 $OUT/pytest_skinny.log
timeout 1800 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --durations=10 > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
grep -E "^E   |^FAILED|passed|failed" $OUT/pytest.log | cut -c1-300 | tail -30
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
rm -f $OUT/ab_batched.jsonl $OUT/ab_batched_oneshot.jsonl
for B in 8 16 32 64; do
  timeout 600 python tools/decode_ab.py --batch $B --prompt-len 256 --steps 128 --profile-steps 4 - >> $OUT/ab_batched.jsonl 2>> $OUT/ab_batched.err
  TL_QMM3_PERSISTENT=0 timeout 600 python tools/decode_ab.py --batch $B --prompt-len 256 --steps 128 --profile-steps 4 - >> $OUT/ab_batched_oneshot.jsonl 2>> $OUT/ab_batched.err
done
python - <<'PY'
import json
for f in ("ab_batched", "ab_batched_oneshot"):
    for l in open(f"gpurun_out/call_f/{f}.jsonl"):
        r=json.loads(l); print(f, "batch",r["batch"],"ms/step",r["ms_per_step"],"tok/s",r["tokens_per_s"],"launches",r.get("launches"))
PY
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -c 600 $OUT/bench.json
timeout 900 python benches/serve_replicas.py --num-seqs 128 --batch-size 64 --json-output $OUT/replicas_n1.json > $OUT/replicas_n1.log 2>&1
echo "replicas rc=$?"; head -12 $OUT/replicas_n1.log
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace_b64 -o b64 --output-format csv -- python $R/tools/decode_ab.py --batch 64 --prompt-len 256 --steps 32 --profile-steps 0 - > $OUT/trace_b64.log 2>&1
echo "trace b64 rc=$?"
cp $R/gpurun_out/parity_numbers.jsonl $OUT/ 2>/dev/null
