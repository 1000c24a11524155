#!/bin/bash
# Full device suite + the driver's bench command (a quick regression pass; the evidence set is tools/evidence_run.sh).
OUT=gpurun_out/quick
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -12 | tee $OUT/tests.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; tail -c 2500 $OUT/bench.json | head -c 1200; echo
python - <<'PY'
import json
b=json.loads(open("gpurun_out/quick/bench.json").read().strip().splitlines()[-1]); r=b["roofline"]
print("value",b["value"],"ms",b["ms_per_step"],"gemv frac",r["frac"],"step_frac",r["step_frac"],"launches",r.get("launches_per_step_all_kernels"),"cpu",b["cpu_baseline"] and b["cpu_baseline"].get("gpu_greedy_ids_vs_truth"))
PY
