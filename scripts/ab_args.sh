#!/bin/bash
# A/B of bench.py argument sets on the GPU box: scripts/ab_args.sh "--max-leaf 1" "--max-leaf 3" ...
cd "$(dirname "$0")/.."
for v in "$@"; do
  line=$(timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-count $v 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "$v => Msamples/s, ms/step: $line"
done
