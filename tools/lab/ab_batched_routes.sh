mkdir -p gpurun_out
for round in 1 2; do
for b in 8 16 32 64; do
  for aql in 1 0; do
    echo "round $round batch $b TL_AQL=$aql $(TL_AQL=$aql python tools/batch_decode_probe.py --batch $b --context 128 --steps 32 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d.get("replay_route",""))')"
  done
done
done
