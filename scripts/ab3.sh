#!/bin/bash
# like ab2.sh but keeps the library's stderr notes (RTX_SCENE_LDS=1 prints the LDS plan)
cd "$(dirname "$0")/.."
wls="$1"; shift
for wl in $wls; do
  for v in "$@"; do
    e="$v"; [ "$v" = "-" ] && e="RTX_NOP=1"
    env $e timeout -k 10 300 python bench.py --workload $wl --steps 6 --warmup 2 --no-cpu-baseline --no-count --no-pmc --no-extras --frames-in-flight 1 > /tmp/ab3.out 2> /tmp/ab3.err
    line=$(grep '^{' /tmp/ab3.out | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    echo "$wl [$v] => Msamples/s, ms/step: $line   $(grep -h '^\[rtx\]' /tmp/ab3.err | sort -u | tr '\n' ' ')"
  done
done
