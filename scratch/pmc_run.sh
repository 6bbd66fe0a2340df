#!/bin/bash
# usage: pmc_run.sh "<ONLY substring>" tag
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_WAVES" \
           "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_LDS" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" \
           "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum"; do
  i=$((i+1))
  ONLY="$1" REP=1 timeout -k 10 120 rocprofv3 --pmc $set -d $R/gpurun_out/pm_$2_$i -o p --output-format csv -- python3 $R/scratch/conv_bench.py > $R/gpurun_out/pm_$2_$i.log 2>&1 || { echo "pass $i failed"; tail -5 $R/gpurun_out/pm_$2_$i.log; }
done
cd $R && python3 scratch/pmc_multi.py gpurun_out/pm_$2_* > gpurun_out/pm_$2.txt
cat gpurun_out/pm_$2.txt
