#!/bin/bash
# Build a VARIANT of the library beside the in-tree one, for A/B runs on one GPU box (MSFWSI_LIB=ab/libmsfwsi_NAME.so):
#   tools/build_variant.sh NAME "-DMSFWSI_FETCH_FIRST=0 ..."
# Sources are copied to a scratch tree (the in-tree objects are not touched); the .so lands in ab/ (git-ignored, travels
# with gpurun).
set -e
NAME=$1; EXTRA=$2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
T=/tmp/msfwsi_variant_$NAME
rm -rf $T && mkdir -p $T/msf_wsi_amd $T/include $ROOT/ab
cp -r $ROOT/msf_wsi_amd/csrc $T/msf_wsi_amd/csrc
cp $ROOT/include/*.h $T/include/
rm -f $T/msf_wsi_amd/csrc/*.o
make -C $T/msf_wsi_amd/csrc -j4 EXTRA="$EXTRA" LIB=$ROOT/ab/libmsfwsi_$NAME.so > $T/build.log 2>&1 || { tail -20 $T/build.log; exit 1; }
ls -la $ROOT/ab/libmsfwsi_$NAME.so
