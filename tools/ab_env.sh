#!/bin/bash
# A/B of engine switches on ONE box, interleaved rounds:  tools/ab_env.sh <rounds> "<ENV=VAL ...>" ["<ENV=VAL ...>" ...]
R=$1; shift
for i in $(seq 1 $R); do
  for E in "default" "$@"; do
    if [ "$E" = "default" ]; then V=""; else V="$E"; fi
    env $V python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-timer 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('[$E]', d['ms_per_step'], d['config']['peak_reserved_GiB'])"
  done
done
