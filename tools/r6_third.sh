#!/bin/bash
O=gpurun_out
timeout -k 10 900 python -m pytest tests/test_options_gpu.py -q -s > $O/r6d_options.log 2>&1; echo "options rc=$?"; tail -12 $O/r6d_options.log | cut -c1-200
timeout -k 10 600 python -m pytest tests/test_blocks_lowp_gpu.py -q -x -k "image" > $O/r6d_blocks.log 2>&1; echo "blocks rc=$?"; tail -2 $O/r6d_blocks.log
echo "== wgrad_bench waves8"; MSFWSI_LIB=$PWD/ab/libmsfwsi_waves8.so timeout -k 10 300 python tools/wgrad_bench.py 2>/dev/null | tee $O/r6d_wgrad_bench_waves8.txt
echo "== wgrad_bench default"; timeout -k 10 300 python tools/wgrad_bench.py 2>/dev/null | tee $O/r6d_wgrad_bench_default.txt
