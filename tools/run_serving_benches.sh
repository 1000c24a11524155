#!/bin/bash
# Reference-shaped workloads beyond the headline bench line (BASELINE.json configs[1..4], one GPU); run via gpurun.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/serving; mkdir -p $OUT; cd $R
run() { name=$1; shift; echo "=== $name: $*"; timeout 600 python -m benches.bench "$@" --json-output $OUT/$name.json 2>&1 | grep -vE "amdgpu.ids"; }
run acceptance_128_129 --num-seqs 1 --min-input-len 128 --max-input-len 128 --min-output-len 129 --max-output-len 129 --warmup 2
#run acceptance_ops --solution ops --num-seqs 1 --min-input-len 128 --max-input-len 128 --min-output-len 33 --max-output-len 33 --warmup 1
run serving_b4 --batch-decode --batch-size 4 --num-seqs 16 --min-input-len 128 --max-input-len 1024 --min-output-len 32 --max-output-len 128 --prefill-step 128
run serving_b8 --batch-decode --batch-size 8 --num-seqs 32 --min-input-len 128 --max-input-len 1024 --min-output-len 32 --max-output-len 128 --prefill-step 128
run serving_b64 --batch-decode --batch-size 64 --num-seqs 128 --min-input-len 128 --max-input-len 1024 --min-output-len 32 --max-output-len 128 --prefill-step 128
run long_8k --num-seqs 1 --min-input-len 8192 --max-input-len 8192 --min-output-len 65 --max-output-len 65 --prefill-step 2048 --warmup 1
run long_32k --num-seqs 1 --min-input-len 32768 --max-input-len 32768 --min-output-len 33 --max-output-len 33 --prefill-step 2048 --warmup 0
