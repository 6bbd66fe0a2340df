#!/bin/bash
# GPU box: per-kernel PMC counters of micro-benchmark launches (tools/kbench.py), one rocprofv3 --pmc pass per counter set:
#   usage: tools/profile_kbench.sh <tag> <kbench cases...>        (env: KBENCH_PANEL_H, KBENCH_ITERS, MSFWSI_LIB ...)
# -> gpurun_out/<tag>_kbench_pmc.txt: per kernel symbol, mean per dispatch of every counter plus derived figures
#    (HBM-side bytes with the gfx950 FETCH correction, L2 hit rate, MFMA utilisation, wait fractions, LDS conflicts).
# python3 stands directly after `--` (the profiler's preload initialises the GPU: no exec hop allowed).
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
export KBENCH_ITERS=${KBENCH_ITERS:-2}
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  rm -rf $OUT/prof_${TAG}_kb$i
  timeout -k 10 600 rocprofv3 --pmc $set -d $OUT/prof_${TAG}_kb$i -o p --output-format csv -- \
      python3 $R/tools/kbench.py "$@" > $OUT/prof_${TAG}_kb$i.log 2>&1 || { echo "pmc pass $i ($set) failed"; tail -5 $OUT/prof_${TAG}_kb$i.log; }
done
cd $R && python3 tools/kbench_pmc_summary.py $OUT/prof_${TAG}_kb? > $OUT/${TAG}_kbench_pmc.txt
rm -rf $OUT/prof_${TAG}_kb?
cat $OUT/${TAG}_kbench_pmc.txt
