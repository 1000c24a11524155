#!/bin/bash
# On the GPU box: HBM bytes of the ENGINE's decode step by hardware counters (tools/lab/engine_step_lab: C ABI only, no Python in the process).
# Separate rocprofv3 --pmc passes for FETCH_SIZE and WRITE_SIZE (no trace domain but the kernel trace), each under its own timeout.
# usage: tools/lab/pmc_engine_step.sh <context> <steps> <batch> <tag> [kv pages: 0 = bf16, 1 = FP8 E4M3]
R=$GRAFT_REPO_ROOT; CTX=${1:-128}; STEPS=${2:-10}; B=${3:-1}; TAG=${4:-ctx$CTX}; KV=${5:-0}
OUT=$R/gpurun_out/pmc_engine_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 120 $R/tools/lab/engine_step_lab $CTX 100 $B $KV > $OUT/plain.log 2>&1; tail -3 $OUT/plain.log
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch -o p --output-format csv -- $R/tools/lab/engine_step_lab $CTX $STEPS $B $KV > $OUT/fetch.log 2>&1 || echo "FETCH_SIZE pass failed"
tail -2 $OUT/fetch.log
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/write -o p --output-format csv -- $R/tools/lab/engine_step_lab $CTX $STEPS $B $KV > $OUT/write.log 2>&1 || echo "WRITE_SIZE pass failed"
tail -2 $OUT/write.log
# the per-dispatch rows are what is needed; the traces are large
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
python3 $R/tools/make_engine_traffic_json.py $OUT $CTX $STEPS $B | tail -40
