#!/bin/bash
# scripts/sweep_env.sh <workload> <VAR> <v1> <v2> ...   -- bench one workload under several values of one tuning variable
wl=$1; var=$2; shift 2
for v in "$@"; do
  export $var=$v
  line=$(timeout -k 10 200 python3 bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-pmc --no-count 2>/dev/null | tail -1)
  echo "$var=$v $(echo "$line" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["roofline"]["kernel_ms"])')"
done
