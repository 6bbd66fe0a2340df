#!/bin/bash
# GPU box, round 6 first contact: the new tests, the weight-gradient A/B (pipelined loop vs the round-5 loop), a bench line
O=gpurun_out
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_headline_geometry_gpu.py tests/test_dist_gpu.py tests/test_train_gpu.py -x -q -s > $O/r6a_tests_new.log 2>&1; echo "new tests rc=$?"; tail -5 $O/r6a_tests_new.log
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_production_gpu.py -x -q -k "wgrad or gram" > $O/r6a_tests_wgrad.log 2>&1; echo "wgrad tests rc=$?"; tail -3 $O/r6a_tests_wgrad.log
echo "== wgrad_bench default (pipelined loop)"; timeout -k 10 300 python tools/wgrad_bench.py 2>/dev/null | tee $O/r6a_wgrad_bench_pipe1.txt
echo "== wgrad_bench pipe0 (round-5 loop)"; MSFWSI_LIB=$PWD/ab/libmsfwsi_pipe0.so timeout -k 10 300 python tools/wgrad_bench.py 2>/dev/null | tee $O/r6a_wgrad_bench_pipe0.txt
timeout -k 10 400 python bench.py --steps 6 --warmup 3 --no-cpu-baseline > $O/r6a_bench.json 2>$O/r6a_bench.err; echo "bench rc=$?"; cut -c1-400 $O/r6a_bench.json
MSFWSI_LIB=$PWD/ab/libmsfwsi_pipe0.so timeout -k 10 400 python bench.py --steps 6 --warmup 3 --no-cpu-baseline > $O/r6a_bench_pipe0.json 2>/dev/null; cut -c1-200 $O/r6a_bench_pipe0.json
