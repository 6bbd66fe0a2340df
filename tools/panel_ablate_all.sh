#!/bin/bash
# GPU box: the panel kernel's default build against ablation builds in ab/ on EVERY panel shape (tools/kbench.py panel)
for lib in default $(ls ab/libmsfwsi_*.so 2>/dev/null); do
  if [ "$lib" = default ]; then unset MSFWSI_LIB; else export MSFWSI_LIB=$PWD/$lib; fi
  echo "== $lib"
  python tools/kbench.py panel 2>/dev/null | grep "^panel"
done
