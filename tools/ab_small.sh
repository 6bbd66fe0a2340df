#!/bin/bash
# GPU box: A/B of the round-5 switches at small batch sizes (bench.py ms/step per environment, one box, two rounds)
run() { echo -n "[$1] "; env $1 timeout -k 10 300 python bench.py $2 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
for shape in "--arch resnet50 --batch 16" "--arch resnet18 --batch 8" "--arch resnet18 --batch 32"; do
  echo "== $shape"
  for round in 1 2; do
    run "X=1" "$shape"
    run "MSFWSI_ENGINE=img3x3=0" "$shape"
    run "MSFWSI_ENGINE=img3x3_min_fill=0" "$shape"
    run "MSFWSI_ENGINE=img3x3_min_fill=2" "$shape"
    run "MSFWSI_ENGINE=heads_on_streams=0" "$shape"
  done
done
