#!/bin/bash
# A/B of the wavefront integrator on one workload (GPU box): scripts/wf_ab.sh <workload> "ENV..." "ENV..." ...
cd "$(dirname "$0")/.."
wl=$1; shift
for v in "$@"; do
  line=$(env $v timeout -k 10 300 python bench.py --workload $wl ${SPP:+--spp $SPP} --steps 3 --warmup 1 --no-cpu-baseline --no-count --no-pmc --no-extras 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline'].get('kernel'))")
  echo "$wl [$v] => Msamples/s, ms/step, kernel: $line"
done
