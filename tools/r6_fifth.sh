#!/bin/bash
O=gpurun_out
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "dgrad2" > $O/r6h_tests_dgrad2.log 2>&1; echo "dgrad2 tests rc=$?"; tail -3 $O/r6h_tests_dgrad2.log
timeout -k 10 900 python -m pytest tests/test_options_gpu.py tests/test_headline_geometry_gpu.py tests/test_blocks_lowp_gpu.py tests/test_lowp_parity_gpu.py -x -q > $O/r6h_tests_model.log 2>&1; echo "model tests rc=$?"; tail -3 $O/r6h_tests_model.log
echo "== kbench dgrad2pro"; timeout -k 10 300 python tools/kbench.py dgrad2pro 2>/dev/null | tee $O/r6h_kbench_dgrad2pro.txt
for i in 1 2; do
timeout -k 10 400 python bench.py --steps 6 --warmup 3 --no-cpu-baseline > $O/r6h_bench_$i.json 2>/dev/null; cut -c1-200 $O/r6h_bench_$i.json
MSFWSI_ENGINE=dgrad2_pro=0 timeout -k 10 400 python bench.py --steps 6 --warmup 3 --no-cpu-baseline > $O/r6h_bench_off_$i.json 2>/dev/null; cut -c1-200 $O/r6h_bench_off_$i.json
done
