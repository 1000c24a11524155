# same-box A/B on the AQL route: attention windows of the batched step (default planner against pinned window sizes)
for round in 1 2; do
for b in 12 16 24 32; do
 for ctx in 150 600; do
  for v in auto 64 128 256; do
    if [ $v = auto ]; then pre=""; else pre="TL_ATTN_MIN_TOKENS=$v"; fi
    echo "round $round batch $b ctx $ctx windows $v $(env $pre python tools/batch_decode_probe.py --batch $b --context $ctx --steps 24 --profile 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "splits", d["profile"]["n_splits"])')"
  done
 done
done
done
