#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/call_b
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
rm -f $R/gpurun_out/parity_numbers.jsonl
timeout 300 python tools/debug_qmv3.py > $OUT/debug_qmv3.log 2>&1
echo "debug rc=$?"; tail -45 $OUT/debug_qmv3.log
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
grep -E "^E   |^FAILED|passed|failed" $OUT/pytest.log | cut -c1-300 | tail -40
timeout 600 python tools/decode_ab.py --batch 8 --prompt-len 256 --steps 128 - TL_QMM3_FUSED_NORM=0 TL_ATTN_RQ=1 > $OUT/ab_b8.jsonl 2> $OUT/ab_b8.err
echo "ab8 rc=$?"; cat $OUT/ab_b8.jsonl
timeout 600 python tools/decode_ab.py --batch 64 --prompt-len 256 --steps 64 - TL_QMM3_FUSED_NORM=0 TL_ATTN_RQ=1 > $OUT/ab_b64.jsonl 2> $OUT/ab_b64.err
echo "ab64 rc=$?"; cat $OUT/ab_b64.jsonl
timeout 600 python tools/decode_ab.py --batch 16 --prompt-len 256 --steps 64 - TL_QMM3_FUSED_NORM=0 > $OUT/ab_b16.jsonl 2> $OUT/ab_b16.err
cat $OUT/ab_b16.jsonl
timeout 900 python bench.py --config 3 --no-cpu-baseline > $OUT/bench_c3.json 2> $OUT/bench_c3.err
echo "c3 rc=$?"; tail -c 2500 $OUT/bench_c3.json
timeout 900 python bench.py --config 5 --no-cpu-baseline > $OUT/bench_c5.json 2> $OUT/bench_c5.err
echo "c5 rc=$?"; tail -c 2500 $OUT/bench_c5.json
cp $R/gpurun_out/parity_numbers.jsonl $OUT/ 2>/dev/null
