#!/bin/bash
# usage: pmc_run2.sh "<ONLY substring>" tag  (instruction-mix counters)
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_BRANCH GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  ONLY="$1" REP=1 timeout -k 10 120 rocprofv3 --pmc $set -d $R/gpurun_out/pi_$2_$i -o p --output-format csv -- python3 $R/scratch/conv_bench.py > $R/gpurun_out/pi_$2_$i.log 2>&1 || { echo "pass $i failed"; tail -5 $R/gpurun_out/pi_$2_$i.log; }
done
cd $R && python3 scratch/pmc_multi.py gpurun_out/pi_$2_* > gpurun_out/pi_$2.txt
