#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/call_c
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
rm -f $R/gpurun_out/parity_numbers.jsonl
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
grep -E "^E   |^FAILED|passed|failed" $OUT/pytest.log | cut -c1-300 | tail -30
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$?"; tail -c 1500 $OUT/bench.json
timeout 600 python benches/bench_week2_operators.py --json-output $OUT/operators.json > $OUT/operators.log 2>&1
echo "ops rc=$?"; cat $OUT/operators.log | tail -14
timeout 600 python benches/bench_week3_attention.py --json-output $OUT/attention.json > $OUT/attention.log 2>&1
echo "attn rc=$?"; cat $OUT/attention.log | tail -10
timeout 600 python bench.py --config 3 --no-cpu-baseline > $OUT/bench_c3.json 2> $OUT/bench_c3.err
python - <<'PY'
import json,sys
for c in ("bench_c3",):
    try:
        b=json.loads(open(f"gpurun_out/call_c/{c}.json").read().strip().splitlines()[-1]); r=b["roofline"]
        print(c,b["value"],b["ms_per_step"],{k:v["us_per_step"] for k,v in r["per_kind"].items()})
    except Exception as e: print(c,"failed",e)
PY
timeout 600 python bench.py --config 5 --no-cpu-baseline > $OUT/bench_c5.json 2> $OUT/bench_c5.err
python - <<'PY'
import json
try:
    b=json.loads(open("gpurun_out/call_c/bench_c5.json").read().strip().splitlines()[-1]); r=b["roofline"]
    print("c5",b["value"],b["ms_per_step"],{k:v["us_per_step"] for k,v in r["per_kind"].items()})
except Exception as e: print("c5 failed",e)
PY
# rocprofv3 kernel trace of the headline command (same flags as r01's committed summary)
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench --output-format csv -- python $R/bench.py --steps 64 --warmup 8 --no-cpu-baseline --profile-steps 0 > $OUT/trace.log 2>&1
echo "trace rc=$?"; find $OUT/trace -name "*kernel_stats.csv" | head -3
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f" | cut -c1-200
cd $R && bash tools/lab/pmc_gemv.sh > $OUT/pmc.log 2>&1; tail -5 $OUT/pmc.log
cp $R/gpurun_out/parity_numbers.jsonl $OUT/ 2>/dev/null
