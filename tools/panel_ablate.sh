#!/bin/bash
# GPU box: the panel kernel's ablation / variant builds (tools/build_variant.sh NAME "-DMSFWSI_PANEL_ABLATE=N") on one shape,
# with the gather kernel's launches of the same shape (case `wide`) first as the box's reference.
# usage: tools/panel_ablate.sh [H list]     (bits: 1 no stores, 2 no MFMAs, 4 no weight reloads, 8 no epilogue operands, 16 no
#   staging loads, 32 no epilogue arithmetic, 64 no gate bits, 128 epilogue accesses as 8 rows x 128 B)
export KBENCH_PANEL_H=${1:-14}
python tools/kbench.py wide 2>/dev/null | grep "^wide ${KBENCH_PANEL_H}x"
for lib in default $(ls ab/libmsfwsi_*.so 2>/dev/null); do
  if [ "$lib" = default ]; then unset MSFWSI_LIB; else export MSFWSI_LIB=$PWD/$lib; fi
  echo "== $lib"
  python tools/kbench.py panel 2>/dev/null | grep "^panel"
done
