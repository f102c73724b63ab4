#!/bin/bash
# bench.py argument sets with the counting run on: prints value, box tests and primitive tests per ray
cd "$(dirname "$0")/.."
for v in "$@"; do
  line=$(timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline $v 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], r['box_tests_per_ray'], r['per_ray'])")
  echo "$v => $line"
done
