#!/bin/bash
# Runs the reference's OWN scripts, unmodified, through the import facade in the build container (no GPU: the numpy oracle
# answers libtinyllm_hip.so's C ABI, tests/run_reference_script.py) on the synthetic stand-in checkpoints, and writes what
# they print to profiles/r02_labs/reference_harness_through_facade.txt.  Evidence that the harness runs end to end on the
# facade + host mirror; the timings are ORACLE timings and mean nothing.
set -u
cd "$(dirname "$0")/.."
export HF_HOME=$(mktemp -d) HF_HUB_OFFLINE=1 OMP_NUM_THREADS=4
python - <<'PY'
import os, sys
sys.path[:0] = [".", "tests"]
from pathlib import Path
from checkpoint_fixture import write_stand_in_checkpoints
write_stand_in_checkpoints(Path(os.environ["HF_HOME"]), eos_friendly=True)
PY
OUT=profiles/r02_labs/reference_harness_through_facade.txt
R="timeout 900 python tests/run_reference_script.py"
C="--model qwen3-0.6b --num-seqs 4 --min-input-len 5 --max-input-len 40 --min-output-len 3 --max-output-len 6 --warmup 1"
S="--model qwen3-0.6b --solution ref --num-seqs 4 --batch-size 2 --min-input-len 10 --max-input-len 40 --min-output-len 2 --max-output-len 4 --warmup 0 --repeats 2 --offline"
run() { echo; echo "\$ python $*"; $R "$@" 2>&1 | grep -v "it/s\]" | tail -40 | cut -c1-220; echo "[exit ${PIPESTATUS[0]}]"; }
{
echo "# reference scripts (from /root/reference, unmodified) through tiny-llm_amd/compat; CPU container, numpy oracle behind the C ABI"
echo "# checkpoints: synthetic stand-ins (tests/checkpoint_fixture.py) under the repository names the scripts look up"
run benches/bench.py $C --solution ref --loader week2
run benches/bench.py $C --solution ref --loader week3 --batch-decode --batch-size 3 --prefill-step 16
run benches/bench.py $C --solution ref --loader week3 --disable-paged-attention --batch-decode --batch-size 3 --prefill-step 16
run benches/bench.py $C --solution ref --loader week2 --batch-decode --batch-size 3 --prefill-step 16
run benches/bench.py $C --solution ref --loader week1 --device cpu
run benches/bench.py $C --solution mlx --device cpu
for c in kv-cache quantized-matvec decode-attention rmsnorm rope swiglu simd-matmul split-k; do
  run benches/bench.py $C --solution ref --loader week2 --week2-checkpoint $c --prefill-logits last
done
run benches/bench_week2_operators.py --model qwen3-0.6b --solution tiny_llm_ref --warmup 1 --iterations 6 --include-split-k
run benches/bench_week3_attention.py --contexts 128 256 --page-size 64 --warmup 1 --iterations 2 --repeats 2
run benches/bench_long_context_attention.py --contexts 128 --warmup 1 --iterations 2 --repeats 1
run benches/profile_week2_kernels.py --model qwen3-0.6b --warmup 1 --iterations 2
run benches/bench_chunked_prefill.py --prefill-steps 8 32 $S
run benches/bench_serving_progression.py --prefill-step 16 $S
run benches/bench_course_progression.py --model qwen3-0.6b --solution ref --input-len 12 --output-len 4 --warmup 0 --repeats 2 --offline
run benches/bench_course_progression.py --suite week2 --model qwen3-0.6b --solution ref --input-len 12 --output-len 4 --warmup 0 --repeats 2 --offline
for a in "--loader week1 --device cpu" "--loader week2" "--loader week3" "--loader week3 --disable-paged-attention" "--loader week2 --draft-model qwen3-8b" "--loader week3 --draft-model qwen3-8b" "--loader week2 --sampler-temp 0.8 --sampler-top-k 5"; do
  run main.py --model qwen3-8b --prompt "w1 w2 w3" --solution ref $a
done
run main.py --model qwen3-8b --prompt "w1 w2 w3" --solution mlx --device cpu
run batch-main.py --model qwen3-8b --solution ref --loader week3 --batch-size 4 --prefill-step 64 --max-seq-len 176
run batch-main.py --model qwen3-8b --solution ref --loader week2 --batch-size 4 --prefill-step 64 --max-seq-len 176
} > $OUT 2>&1
echo "scripts that exited 0: $(grep -c "^\[exit 0\]" $OUT)"; grep "^\[exit [^0]" $OUT || true
