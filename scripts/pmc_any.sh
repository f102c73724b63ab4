#!/bin/bash
# One rocprofv3 counter pass over bench.py for a kernel variant; prints the mean of each counter over the trace kernel's launches.
#   scripts/pmc_any.sh <tag> "<COUNTER ...>" ENV=... ENV=...
tag=$1; shift
counters=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
env "$@" rocprofv3 --kernel-trace --pmc $counters --output-format csv -d $R/gpurun_out/pmc_$tag -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-count > $R/gpurun_out/pmc_$tag.log 2>&1
python3 - <<PY
import csv,glob,collections
f=glob.glob("$R/gpurun_out/pmc_$tag/*/*_counter_collection.csv")[0]
agg=collections.defaultdict(list); name=""
for r in csv.DictReader(open(f)):
    if "k_trace" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"])); name=r["Kernel_Name"][:48]
print("$tag", name)
for k,v in sorted(agg.items()): print("  %-28s %.4e" % (k, sum(v)/len(v)))
PY
