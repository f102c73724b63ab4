#!/bin/bash
# A/B on the GPU box: one bench.py process per (workload, variant); env is read at scene upload / flatten.
# usage: scripts/ab2.sh "c2 head" "ENV1=a ENV2=b" "ENV1=c" ...   ("-" = no environment change)
cd "$(dirname "$0")/.."
wls="$1"; shift
for wl in $wls; do
  for v in "$@"; do
    e="$v"; [ "$v" = "-" ] && e="RTX_NOP=1"
    line=$(env $e timeout -k 10 300 python bench.py --workload $wl --steps 6 --warmup 2 --no-cpu-baseline --no-count --no-pmc --no-extras --frames-in-flight 1 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    echo "$wl [$v] => Msamples/s, ms/step: $line"
  done
done
