#!/bin/bash
# GPU box: rocprofv3 evidence for one bench configuration.   usage: tools/profile_step.sh <tag> [bench args...]
#   1. --kernel-trace --stats of `python3 bench.py --steps 4 --warmup 2`   -> gpurun_out/<tag>_kernel_stats.csv
#   2. separate --pmc passes of `python3 bench.py --steps 1 --warmup 1`:
#        FETCH_SIZE | WRITE_SIZE                                 (HBM-side traffic; TCC slots do not fit both)
#        SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE   (MFMA utilisation)
#        SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY (LDS conflicts, stalls)
#      -> gpurun_out/<tag>_pmc.json via tools/pmc_summary.py
# python3 stands directly after `--` (the profiler's preload initialises the GPU: no exec hop allowed).
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
ARGS="--no-cpu-baseline $*"
rm -rf $OUT/prof_${TAG}_kt
timeout -k 10 900 rocprofv3 --kernel-trace --stats -d $OUT/prof_${TAG}_kt -o kt --output-format csv -- \
    python3 $R/bench.py --steps 4 --warmup 2 $ARGS > $OUT/prof_${TAG}_kt.log 2>&1 || { echo "kernel-trace pass failed"; tail -5 $OUT/prof_${TAG}_kt.log; exit 1; }
cp $(find $OUT/prof_${TAG}_kt -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_kernel_stats.csv
if [ "${KT_ONLY:-0}" = "1" ]; then rm -rf $OUT/prof_${TAG}_kt; exit 0; fi   # kernel trace only (e.g. a second stream mode)
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rm -rf $OUT/prof_${TAG}_pmc$i
  timeout -k 10 900 rocprofv3 --pmc $set -d $OUT/prof_${TAG}_pmc$i -o p --output-format csv -- \
      python3 $R/bench.py --steps 1 --warmup 1 --no-kernel-timer $ARGS > $OUT/prof_${TAG}_pmc$i.log 2>&1 || { echo "pmc pass $i failed"; tail -5 $OUT/prof_${TAG}_pmc$i.log; exit 1; }
done
cd $R && python3 tools/pmc_summary.py $OUT/prof_${TAG}_pmc1 $OUT/prof_${TAG}_pmc2 $OUT/prof_${TAG}_pmc3 $OUT/prof_${TAG}_pmc4 \
    $OUT/${TAG}_kernel_stats.csv $OUT/${TAG}_pmc.json
# the raw per-dispatch counter CSVs are large: keep only the summaries
rm -rf $OUT/prof_${TAG}_pmc1 $OUT/prof_${TAG}_pmc2 $OUT/prof_${TAG}_pmc3 $OUT/prof_${TAG}_pmc4 $OUT/prof_${TAG}_kt
