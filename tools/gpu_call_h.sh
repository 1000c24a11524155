#!/bin/bash
# r02 call H: packed multi-sequence prefill -- parity test, serving A/B (one staging slot vs 16) on the config-4 trace.
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/call_h
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_engine_gpu.py -m gpu -q --tb=short -p no:cacheprovider -x -k "packed or continuous or prefill" > $OUT/pytest_packed.log 2>&1
echo "pytest packed rc=$?"; tail -12 $OUT/pytest_packed.log | cut -c1-400
for S in 1 16; do
  timeout 900 python benches/serve_replicas.py --num-seqs 128 --batch-size 64 --staging-slots $S --json-output $OUT/replicas_s$S.json > $OUT/replicas_s$S.log 2>&1
  echo "replicas staging $S rc=$?"; grep -E "^Time|^Total|^Prefill|^Decode throughput|Decode step p50|Peak active" $OUT/replicas_s$S.log
done
timeout 900 python benches/serve_replicas.py --num-seqs 128 --batch-size 64 --staging-slots 8 --prefill-step 512 --json-output $OUT/replicas_s8_512.json > $OUT/replicas_s8_512.log 2>&1
echo "replicas staging 8 step 512 rc=$?"; grep -E "^Time|^Total|^Prefill|^Decode throughput|Decode step p50|Peak active" $OUT/replicas_s8_512.log
