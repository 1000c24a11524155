#!/bin/bash
# On the GPU box: batched decode steps on another build of the library (_ab_old/extensions_hip, not tracked) and on the current tree, alternating, same box.
# usage: tools/r6_ab_batched.sh "<batches>" [rounds]
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/ab_batched.jsonl; : > $OUT
for i in $(seq 1 ${2:-3}); do
  for which in old new; do
    if [ $which = old ]; then export TL_EXT_ROOT=$R/_ab_old/extensions_hip; else unset TL_EXT_ROOT; fi
    for b in ${1:-8 16 32 64}; do
      python $R/tools/batch_decode_probe.py --batch $b --context 128 --steps 64 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(json.dumps({'which':'$which','batch':$b,'ms':d.get('ms_per_step')}))" >> $OUT
    done
  done
done
python - <<PY
import json,collections
d=collections.defaultdict(list)
for l in open("$OUT"):
    r=json.loads(l); d[(r['batch'],r['which'])].append(r['ms'])
for (b,w),v in sorted(d.items()): print(b,w,' '.join('%.4f'%x for x in v),'min %.4f'%min(v))
PY
