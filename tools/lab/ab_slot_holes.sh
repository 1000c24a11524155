R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/serving_ab; mkdir -p $OUT; cd $R
run() { name=$1; shift; echo "=== $name: $*"; timeout 300 python -m benches.bench "$@" --json-output $OUT/$name.json 2>&1 | grep -E "Total throughput|Decode step latency|Decode throughput|Time:"; }
B64="--batch-decode --batch-size 64 --num-seqs 128 --min-input-len 128 --max-input-len 1024 --min-output-len 32 --max-output-len 128 --prefill-step 128"
run b64_holes_kept $B64 --keep-slot-holes
run b64_holes_closed $B64
run b64_holes_kept_2 $B64 --keep-slot-holes
run b64_holes_closed_2 $B64
run b8_holes_closed --batch-decode --batch-size 8 --num-seqs 32 --min-input-len 128 --max-input-len 1024 --min-output-len 32 --max-output-len 128 --prefill-step 128
echo "=== replicas budget 2048 / 8 staging"; for f in "" "--keep-slot-holes"; do timeout 300 python -m benches.serve_replicas --num-seqs 128 --batch-size 64 --gpus 1 $f --json-output $OUT/replicas$f.json 2>&1 | grep -E "Total throughput|Decode step p50"; done
