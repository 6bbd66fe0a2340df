#!/bin/bash
# GPU box: one kbench case on the default build and on every variant build in ab/ (tools/build_variant.sh).  usage: tools/ab_kbench.sh CASE [grep pattern]
for lib in default $(ls ab/libmsfwsi_*.so 2>/dev/null); do
  if [ "$lib" = default ]; then unset MSFWSI_LIB; else export MSFWSI_LIB=$PWD/$lib; fi
  echo "== $lib"
  python tools/kbench.py $1 2>/dev/null | grep "${2:-.}"
done
