#!/bin/bash
# On the GPU box: chunked prefill on another build of the library (_ab_old/extensions_hip, not tracked) and on the current tree, alternating, same box.
# usage: tools/r6_ab_prefill.sh [rounds]
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/ab_prefill.jsonl; : > $OUT
for i in $(seq 1 ${1:-2}); do
  for which in old new; do
    if [ $which = old ]; then export TL_EXT_ROOT=$R/_ab_old/extensions_hip; else unset TL_EXT_ROOT; fi
    for pc in "8192 4096" "32768 4096" "8192 2048"; do
      set -- $pc
      python $R/tools/prefill_probe.py --prompt $1 --chunk $2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(json.dumps({'which':'$which','prompt':$1,'chunk':$2,'tokens_per_s':d.get('tokens_per_s')}))" >> $OUT
    done
  done
done
cat $OUT
