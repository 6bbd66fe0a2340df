#!/bin/bash
# determinism / uninitialised-memory sweep:  tools/race_sweep.sh lib...   ("default" = the in-tree build)
cd "$(dirname "$0")/.."
for lib in "$@"; do
  if [ "$lib" = default ]; then unset MSFWSI_LIB; else export MSFWSI_LIB="$PWD/$lib"; fi
  timeout -k 10 200 python -u tools/race_check.py resnet18 8 64 bf16 6 train-poison || exit 1
  timeout -k 10 200 python -u tools/race_check.py resnet50 8 64 fp32 8 train-poison || exit 1
  timeout -k 10 200 python -u tools/race_check.py resnet50 8 64 bf16 10 train-poison || exit 1
  MSFWSI_TUNING=15=1 timeout -k 10 200 python -u tools/race_check.py resnet18 16 64 bf16 8 train || exit 1
  MSFWSI_TUNING=15=1 timeout -k 10 200 python -u tools/race_check.py resnet50 8 64 bf16 8 train || exit 1
  MSFWSI_TUNING=15=1 timeout -k 10 200 python -u tools/race_check.py resnet50 8 64 fp32 8 train || exit 1
  # the fused trainer on three streams (round 4): 4 steps per repetition, bitwise against the first repetition
  MSFWSI_TUNING=15=1 timeout -k 10 300 python -u tools/race_check.py resnet18 16 64 bf16 6 trainer-poison || exit 1
  MSFWSI_TUNING=15=1 timeout -k 10 300 python -u tools/race_check.py resnet18 16 64 fp32 4 trainer-poison || exit 1
  MSFWSI_TUNING=15=1 timeout -k 10 300 python -u tools/race_check.py resnet18 16 64 bf16 6 trainer-poison || exit 1
  MSFWSI_TUNING=15=1 timeout -k 10 600 python -u tools/race_check.py resnet50 8 64 bf16 3 trainer-poison || exit 1
done
