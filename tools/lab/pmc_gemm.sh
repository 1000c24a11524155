#!/bin/bash
# On the GPU box: hardware counters of the prefill GEMM lab (one rocprofv3 pass per counter set; --pmc passes carry no trace domains but the kernel trace).
# usage: tools/lab/pmc_gemm.sh <lab binary> <rows> <tag>
BIN=$1; M=$2; TAG=$3
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_gemm_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
  "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD" \
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM" \
  "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum" \
  "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $set --kernel-trace -d $OUT/p$i -o p --output-format csv -- $GRAFT_REPO_ROOT/$BIN $M 2 > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
python3 $GRAFT_REPO_ROOT/tools/lab/pmc_gemm_join.py $OUT > $OUT/summary.txt 2>&1; cat $OUT/summary.txt
