#!/bin/bash
O=gpurun_out
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_production_gpu.py -x -q -k "wgrad or gram" > $O/r6c_tests_wgrad.log 2>&1; echo "wgrad tests rc=$?"; tail -2 $O/r6c_tests_wgrad.log
echo "== wgrad_bench default (4 stages, requests before the barrier)"; timeout -k 10 300 python tools/wgrad_bench.py 2>/dev/null | tee $O/r6c_wgrad_bench_early.txt
echo "== wgrad_bench stages3 (pipelined, requests behind the barrier)"; MSFWSI_LIB=$PWD/ab/libmsfwsi_stages3.so timeout -k 10 300 python tools/wgrad_bench.py 2>/dev/null | tee $O/r6c_wgrad_bench_stages3.txt
timeout -k 10 800 python -m pytest tests/test_headline_geometry_gpu.py -q -s > $O/r6c_headline.log 2>&1; echo "headline rc=$?"; tail -3 $O/r6c_headline.log
for i in 1 2; do
timeout -k 10 400 python bench.py --steps 6 --warmup 3 --no-cpu-baseline > $O/r6c_bench_$i.json 2>/dev/null; cut -c1-200 $O/r6c_bench_$i.json
MSFWSI_LIB=$PWD/ab/libmsfwsi_pipe0.so timeout -k 10 400 python bench.py --steps 6 --warmup 3 --no-cpu-baseline > $O/r6c_bench_pipe0_$i.json 2>/dev/null; cut -c1-200 $O/r6c_bench_pipe0_$i.json
done
