#!/bin/bash
# GPU box: A/B of several builds of the library on ONE box, interleaved rounds.   usage: tools/ab_libs.sh ROUNDS lib1.so lib2.so ...
# ("default" = the in-tree msf_wsi_amd/libmsfwsi_hip.so).  Prints ms/step of `bench.py --steps 5 --warmup 2` per build and round.
R=$1; shift
for r in $(seq 1 $R); do
  for lib in "$@"; do
    if [ "$lib" = default ]; then unset MSFWSI_LIB; else export MSFWSI_LIB=$PWD/$lib; fi
    python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timer 2>/dev/null | grep -o "\"ms_per_step\": [0-9.]*" | sed "s|^|round $r $lib |"
  done
done
