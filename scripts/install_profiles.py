"""Copy one scripts/profile_round.sh result directory (under gpurun_out/) into profiles/<round>/, keeping only the
rows of this library's kernels in the per-dispatch counter CSVs.

    python scripts/install_profiles.py gpurun_out/prof_final profiles/r01 [full_bench_line.log]
"""
import csv
import json
import os
import shutil
import sys

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, "bench_c2.json"))
shutil.copy(os.path.join(src, "kernel_stats.csv"), os.path.join(dst, "bench_c2_kernel_stats.csv"))
for n in ("fetch_size", "write_size", "sq", "lds"):
    rows = list(csv.reader(open(os.path.join(src, "pmc_%s.csv" % n))))
    ki = rows[0].index("Kernel_Name")
    keep = [rows[0]] + [r for r in rows[1:] if "rtx::" in r[ki]]
    csv.writer(open(os.path.join(dst, "pmc_%s.csv" % n), "w")).writerows(keep)
shutil.copy(os.path.join(src, "pmc_traffic.json"), os.path.join(os.path.dirname(dst.rstrip("/")), "pmc_traffic.json"))
if len(sys.argv) > 3:
    line = [l for l in open(sys.argv[3]) if l.startswith("{")][-1]
    json.loads(line)
    open(os.path.join(dst, "bench_c2_full.json"), "w").write(line)
print("installed", src, "->", dst)
