#!/bin/bash
# GPU box: the image-stationary 3x3 kernel's ablation builds (tools/build_variant.sh img_aN "-DMSFWSI_IMG_ABLATE=N"; bits:
# 1 no staging loads, 2 one tap instead of nine, 4 no output stores / mask loads, 8 no weight re-loads, 16 no LDS reads in
# the k loop) on the two served shapes; the default build first.
for lib in default $(ls ab/libmsfwsi_img_*.so 2>/dev/null); do
  if [ "$lib" = default ]; then unset MSFWSI_LIB; else export MSFWSI_LIB=$PWD/$lib; fi
  echo "== $lib"
  python tools/kbench.py img3 2>/dev/null | grep "image"
done
