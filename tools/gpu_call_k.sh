#!/bin/bash
# r02 call K: batched decode table + serving + 64-sequence trace again (call J landed on a box whose MFMA / LDS-heavy kernels
# ran 1.3-1.6x slower than on the boxes of calls F, H and I with the same binaries; HBM-bound single-stream numbers were normal)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/call_k
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
rocm-smi --showclocks --showpower --showperflevel > $OUT/rocm_smi_before.txt 2>&1
rm -f $OUT/ab_batched.jsonl
for B in 2 4 8 16 32 64; do
  timeout 600 python tools/decode_ab.py --batch $B --prompt-len 256 --steps 128 --profile-steps 4 - >> $OUT/ab_batched.jsonl 2>> $OUT/ab_batched.err
done
python - <<'PY'
import json
for l in open("gpurun_out/call_k/ab_batched.jsonl"):
    r=json.loads(l); print("batch",r["batch"],"ms/step",r["ms_per_step"],"tok/s",r["tokens_per_s"],"launches",r.get("launches"))
PY
timeout 900 python benches/serve_replicas.py --num-seqs 128 --batch-size 64 --json-output $OUT/replicas_n1.json > $OUT/replicas_n1.log 2>&1
echo "replicas rc=$?"; grep -E "^Time|^Total|^Prefill|^Decode throughput|Decode step p50|Peak active" $OUT/replicas_n1.log
timeout 900 python benches/serve_replicas.py --num-seqs 128 --batch-size 64 --staging-slots 1 --prefill-step 128 --json-output $OUT/replicas_n1_reference_admission.json > $OUT/replicas_n1_reference_admission.log 2>&1
grep -E "^Time|^Total|^Prefill|^Decode throughput" $OUT/replicas_n1_reference_admission.log
rocm-smi --showclocks --showpower > $OUT/rocm_smi_after.txt 2>&1
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace_b64 -o b64 --output-format csv -- python $R/tools/decode_ab.py --batch 64 --prompt-len 256 --steps 32 --profile-steps 0 - > $OUT/trace_b64.log 2>&1
echo "trace b64 rc=$?"
grep -E "sclk|mclk|Power" $OUT/rocm_smi_before.txt | head -8
