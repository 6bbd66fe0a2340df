for args in "--arch resnet18 --batch 8" "--arch resnet18 --batch 32" "--arch resnet18 --batch 256" "--arch resnet50 --batch 16" "--arch resnet50 --batch 64" "--config 5"; do
  echo "== $args"
  timeout -k 10 300 python bench.py $args --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['unit'], d['config'].get('recompute_plan'))"
done
