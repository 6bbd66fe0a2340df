#!/bin/bash
# A/B of two builds of the library on ONE box, interleaved rounds:  tools/ab_bench.sh <libA.so> <libB.so> [rounds] [bench args]
A=$1; B=$2; R=${3:-2}; shift 3
for i in $(seq 1 $R); do
  for L in $A $B; do
    MSFWSI_LIB=$L python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-timer "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$L', d['ms_per_step'])"
  done
done
