#!/bin/bash
# All the rocprofv3 evidence for the headline workload, one pass per counter group (gpurun refuses mixed tracing):
#   scripts/profile_round.sh <outdir-under-gpurun_out> [bench args]
# Produces <out>/bench.json (the bench line of the --stats run), <out>/kernel_stats.csv, <out>/pmc_*.csv and
# <out>/pmc_traffic.json (FETCH_SIZE x2 + WRITE_SIZE per trace launch, MI355X_MICROARCH.md HBM section).
out=$1; shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline "$@" > $O/bench_stats.log 2>&1 || exit 1
grep '^{' $O/bench_stats.log > $O/bench.json
cp $(ls $O/stats/*/*_kernel_stats.csv | head -1) $O/kernel_stats.csv
for grp in "fetch_size:FETCH_SIZE" "write_size:WRITE_SIZE" \
           "sq:SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "lds:SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  name=${grp%%:*}; ctr=${grp#*:}
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $O/pmc_$name -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-count "$@" > $O/pmc_$name.log 2>&1 || exit 1
  cp $(ls $O/pmc_$name/*/*_counter_collection.csv | head -1) $O/pmc_$name.csv
done
python3 - <<PY
import csv, json, collections
O="$O"
def mean_counters(path):
    agg=collections.defaultdict(list); name=""
    for r in csv.DictReader(open(path)):
        if "k_trace" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"])); name=r["Kernel_Name"].split("(")[0]
    return {k: sum(v)/len(v) for k,v in agg.items()}, name
f,_=mean_counters(O+"/pmc_fetch_size.csv"); w,_=mean_counters(O+"/pmc_write_size.csv")
sq,name=mean_counters(O+"/pmc_sq.csv"); lds,_=mean_counters(O+"/pmc_lds.csv")
ks=[r for r in csv.DictReader(open(O+"/kernel_stats.csv")) if "k_trace" in r["Name"]]
bench=json.loads(open(O+"/bench.json").read())
rec={"workload": "c2", "spp": bench["config"]["spp"], "kernel": name, "round": 1,
     "fetch_size_kb": f["FETCH_SIZE"], "write_size_kb": w["WRITE_SIZE"],
     "fetch_correction": "x2 (gfx950 wide-read undercount, MI355X_MICROARCH.md HBM section)",
     "hbm_bytes_per_launch": f["FETCH_SIZE"]*1024*2 + w["WRITE_SIZE"]*1024,
     "kernel_trace_mean_ms": float(ks[0]["AverageNs"])/1e6 if ks else None, "kernel_trace_launches": int(ks[0]["Calls"]) if ks else None,
     "bench_kernel_ms": bench["roofline"]["kernel_ms"], "bench_value": bench["value"],
     "sq": sq, "lds_and_issue": lds,
     "lane_utilisation": sq["SQ_THREAD_CYCLES_VALU"]/(64*sq["SQ_ACTIVE_INST_VALU"]),
     "valu_active_per_wave_cycle": sq["SQ_ACTIVE_INST_VALU"]/sq["SQ_WAVE_CYCLES"],
     "source": "rocprofv3 --kernel-trace --pmc <group> (separate passes), scripts/profile_round.sh"}
json.dump(rec, open(O+"/pmc_traffic.json","w"), indent=1)
print(json.dumps(rec, indent=1))
PY
rm -rf $O/stats $O/pmc_fetch_size $O/pmc_write_size $O/pmc_sq $O/pmc_lds
