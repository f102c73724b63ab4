#!/bin/bash
# SQ counter pass over bench.py for one kernel variant: scripts/pmc_sq.sh <tag> ENV=... ENV=...
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
env "$@" rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/pmc_$tag -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-count > $R/gpurun_out/pmc_$tag.log 2>&1
python3 - <<PY
import csv,glob,collections
f=glob.glob("$R/gpurun_out/pmc_$tag/*/*_counter_collection.csv")[0]
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "k_trace" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        vg=r["VGPR_Count"]; name=r["Kernel_Name"][:40]
m={k:sum(v)/len(v) for k,v in agg.items()}
print("$tag", name, "vgpr", vg)
print("  INSTS_VALU %.3e  ACTIVE_INST_VALU %.3e  THREAD_CYCLES_VALU %.3e  WAVE_CYCLES %.3e WAIT_ANY %.3e WAIT_INST_ANY %.3e" % (m["SQ_INSTS_VALU"], m["SQ_ACTIVE_INST_VALU"], m["SQ_THREAD_CYCLES_VALU"], m["SQ_WAVE_CYCLES"], m["SQ_WAIT_ANY"], m["SQ_WAIT_INST_ANY"]))
print("  lane utilisation %.3f   valu-active/wave-cycles %.3f   cycles/inst %.2f" % (m["SQ_THREAD_CYCLES_VALU"]/(64*m["SQ_ACTIVE_INST_VALU"]), m["SQ_ACTIVE_INST_VALU"]/m["SQ_WAVE_CYCLES"], 4*m["SQ_ACTIVE_INST_VALU"]/m["SQ_INSTS_VALU"]))
PY
