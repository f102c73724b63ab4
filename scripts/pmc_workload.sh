#!/bin/bash
# SQ counter pass for one bench workload: scripts/pmc_workload.sh <tag> <bench args...>
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/pmcw_$tag -- python3 $R/bench.py "$@" --steps 1 --warmup 1 --no-cpu-baseline --no-count > $R/gpurun_out/pmcw_$tag.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INSTS_FLAT --output-format csv -d $R/gpurun_out/pmcw2_$tag -- python3 $R/bench.py "$@" --steps 1 --warmup 1 --no-cpu-baseline --no-count > $R/gpurun_out/pmcw2_$tag.log 2>&1
python3 - <<PY
import csv,glob,collections
for d in ("pmcw_$tag","pmcw2_$tag"):
    fs=glob.glob("$R/gpurun_out/%s/*/*_counter_collection.csv"%d)
    if not fs: print(d,"no counters"); continue
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "k_trace" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"])); name=r["Kernel_Name"][:44]; vg=r["VGPR_Count"]; sg=r["SGPR_Count"]
    m={k:sum(v)/len(v) for k,v in agg.items()}
    print(d, name, "vgpr",vg,"sgpr",sg)
    print("  "+"  ".join("%s %.3e"%(k.replace("SQ_",""),v) for k,v in sorted(m.items())))
    if "SQ_THREAD_CYCLES_VALU" in m:
        print("  lane util %.3f  valu-active/wave-cycles %.3f  wait_any/wave-cycles %.3f"%(m["SQ_THREAD_CYCLES_VALU"]/(64*m["SQ_ACTIVE_INST_VALU"]), m["SQ_ACTIVE_INST_VALU"]/m["SQ_WAVE_CYCLES"], m["SQ_WAIT_ANY"]/m["SQ_WAVE_CYCLES"]))
PY
