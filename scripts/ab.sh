#!/bin/bash
# A/B of trace-kernel variants on the GPU box: one bench.py process per variant (env is read at scene upload).
# usage: tools/ab.sh "ENV1=a ENV2=b" "ENV1=c" ...   (each argument = one variant's environment)
cd "$(dirname "$0")/.."
for v in "$@"; do
  line=$(env $v timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-count 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "$v => Msamples/s, ms/step: $line"
done
