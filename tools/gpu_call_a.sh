#!/bin/bash
# GPU box script (via gpurun): full GPU test tier, the default bench line, and the attention A/B.
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/call_a
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
rm -f $R/gpurun_out/parity_numbers.jsonl
timeout 1500 python -m pytest tests -m gpu -q --tb=short --durations=25 -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -60 $OUT/pytest.log
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$?"; tail -c 3000 $OUT/bench.json
timeout 600 python tools/decode_ab.py - TL_ATTN_WIDE_MAX=0 TL_ATTN_NW=8 TL_ATTN_NW=16 TL_ATTN_VECTOR_IDS=1 TL_ATTN_WIDE_MAX=256 > $OUT/ab_attn.jsonl 2> $OUT/ab_attn.err
echo "ab rc=$?"; cat $OUT/ab_attn.jsonl
cp $R/gpurun_out/parity_numbers.jsonl $OUT/ 2>/dev/null
