#!/bin/bash
O=gpurun_out
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_production_gpu.py -x -q -k "conv_fwd or dgrad or big_tile or post or two_source or stem or beyond" > $O/r6f_tests_conv.log 2>&1; echo "conv tests rc=$?"; tail -2 $O/r6f_tests_conv.log
echo "== kbench deep, default (pipelined igemm loop)"; timeout -k 10 300 python tools/kbench.py deep 2>/dev/null | grep -v wgrad | tee $O/r6f_kbench_deep_pipe1.txt
echo "== kbench deep, igpipe0"; MSFWSI_LIB=$PWD/ab/libmsfwsi_igpipe0.so timeout -k 10 300 python tools/kbench.py deep 2>/dev/null | grep -v wgrad | tee $O/r6f_kbench_deep_pipe0.txt
echo "== kbench wide/epi3 default"; timeout -k 10 300 python tools/kbench.py wide dma 2>/dev/null | tee $O/r6f_kbench_wide_pipe1.txt
echo "== kbench wide/epi3 igpipe0"; MSFWSI_LIB=$PWD/ab/libmsfwsi_igpipe0.so timeout -k 10 300 python tools/kbench.py wide dma 2>/dev/null | tee $O/r6f_kbench_wide_pipe0.txt
for i in 1 2; do
timeout -k 10 400 python bench.py --steps 6 --warmup 3 --no-cpu-baseline > $O/r6f_bench_$i.json 2>/dev/null; cut -c1-200 $O/r6f_bench_$i.json
MSFWSI_LIB=$PWD/ab/libmsfwsi_igpipe0.so timeout -k 10 400 python bench.py --steps 6 --warmup 3 --no-cpu-baseline > $O/r6f_bench_igpipe0_$i.json 2>/dev/null; cut -c1-200 $O/r6f_bench_igpipe0_$i.json
done
