#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out
cd /tmp; export TMPDIR=/tmp
rm -rf $O/prof_tl
timeout -k 10 600 rocprofv3 --kernel-trace -d $O/prof_tl -o tl --output-format csv -- python3 $R/bench.py --steps 4 --warmup 3 --no-kernel-timer --no-cpu-baseline > $O/r6_timeline_bench.log 2>&1 || { tail -5 $O/r6_timeline_bench.log; exit 1; }
F=$(find $O/prof_tl -name '*kernel_trace.csv' | head -1)
cd $R && python3 tools/timeline_gaps.py $F 3 | tee $O/r6_timeline_gaps.txt
rm -rf $O/prof_tl
