#!/bin/bash
# One line per workload (f64 and f32) from the default bench's extras: scripts/bench_all.sh <tag>   (GPU box)
cd "$(dirname "$0")/.."
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pmc --no-count > gpurun_out/bench_all_$1.json 2> gpurun_out/bench_all_$1.err
python - "$1" <<'P'
import json, sys
d = json.loads(open("gpurun_out/bench_all_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
print("c2", d["value"], d["ms_per_step"])
for k, v in d.get("other_workloads", {}).items():
    print(k, v["value"], v["ms_per_step"], v["kernel"])
P
