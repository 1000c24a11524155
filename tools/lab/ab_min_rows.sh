# same-box A/B: from how many sequences on does the batched-matmul step beat the fused-GEMV step? (TL_QMM3_MIN_M = first row count of the batched step)
for round in 1 2; do
for b in 2 3 4 5 6; do
  for m in 5 3 2; do
    if [ $m -le $b ] || [ $m -eq 5 ]; then
    echo "round $round batch $b TL_QMM3_MIN_M=$m $(TL_QMM3_MIN_M=$m python tools/batch_decode_probe.py --batch $b --context 128 --steps 32 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])')"
    fi
  done
done
done
