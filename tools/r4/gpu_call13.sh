#!/bin/bash
# round 4, call 13: decode attention with the unconditional stage prefetch -- parity, then timing at short / 8k / 32k contexts and at 64 sequences
OUT=gpurun_out/r4c13
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_decode_kernels_gpu.py -q -k "attention" 2>&1 | tail -4 | tee $OUT/pytest.txt
timeout 600 python -m pytest tests/test_zz_engine_windows_vs_truth_gpu.py tests/test_engine_qwen4b_gpu.py -q 2>&1 | tail -3 | tee -a $OUT/pytest.txt
timeout 300 python bench.py --gpus 1 --steps 40 --warmup 5 --no-cpu-baseline --rocprof off 2> /dev/null | tee $OUT/bench_c2.json | python -c "import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2', b['value'], b['ms_per_step'])"
timeout 600 python bench.py --config 3 --no-cpu-baseline --rocprof off 2> /dev/null | tee $OUT/bench_c3.json | python -c "import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c3', b['value'], b['ms_per_step'], b['roofline']['attention_kv'])"
timeout 900 python bench.py --config 5 --no-cpu-baseline --rocprof off 2> /dev/null | tee $OUT/bench_c5.json | python -c "import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c5', b['value'], b['ms_per_step'], b['roofline']['attention_kv'])"
for b in 64 16; do timeout 300 python tools/decode_ab.py --batch $b --prompt-len 128 --steps 64 - - 2>&1 | grep -v Warning | tee -a $OUT/ab.jsonl; done
echo done
