#!/bin/bash
# On the GPU box: the driver's command (timed region only) on the library of the round's first commit (_ab_old/, built by hand from a
# git worktree; not tracked) and on the current tree, alternating, same box.  usage: tools/r6_ab_old_new.sh [rounds]
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/ab_old_new.jsonl; : > $OUT
for i in $(seq 1 ${1:-3}); do
  for which in old new; do
    if [ $which = old ]; then B=$R/_ab_old/bench.py; else B=$R/bench.py; fi
    python $B --no-cpu-baseline --no-extra-configs --no-clocks --rocprof off 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(json.dumps({'which':'$which','value':d['value'],'ms':d['ms_per_step'],'attention_us':r['per_kind']['attention']['us_per_step'],'kernels_us':r['kernel_time_us_per_step']}))" >> $OUT
  done
done
cat $OUT
