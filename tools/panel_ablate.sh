#!/bin/bash
# GPU box: the panel kernel's ablation builds (tools/build_variant.sh pabN "-DMSFWSI_PANEL_ABLATE=N") on the 14x14 shape.
# usage: tools/panel_ablate.sh [H list]     (bits: 1 no stores, 2 no MFMAs, 4 no weight reloads, 8 no epilogue operands, 16 no staging loads)
export KBENCH_PANEL_H=${1:-14}
for lib in default $(ls ab/libmsfwsi_pab*.so 2>/dev/null); do
  if [ "$lib" = default ]; then unset MSFWSI_LIB; else export MSFWSI_LIB=$PWD/$lib; fi
  echo "== $lib"
  python tools/kbench.py panel 2>/dev/null | grep "^panel"
done
