#!/usr/bin/env python3
"""Which rocprofv3 SQ_INSTS_VALU_* class counter does each VALU opcode of the shipped kernels tick?

    opcode_classes.py <classify_a.csv> <classify_b.csv> <issue_calib.json> <out.json>

The CSVs are rocprofv3 --pmc collections over lib/issue_calib (scripts/pmc_classify.sh): one dispatch per opcode
(kernel k_auto<i>), each executing a known number of wave-instructions of that one opcode.  Output:
{"class_of": {mnemonic: "FMA_F32" | ... | "OTHER"}, "valu_per_inst": {...}, "active_quads_per_inst": {...}}.
An opcode belongs to a class when that class counter ticks >= 0.5 per executed wave-instruction.
"""
import collections
import csv
import json
import re
import sys


def load(path):
    rows = collections.defaultdict(dict)
    for r in csv.DictReader(open(path)):
        m = re.search(r"k_auto<(\d+)>", r["Kernel_Name"])
        if m:
            rows[int(m.group(1))][r["Counter_Name"]] = float(r["Counter_Value"])
    return rows


def main():
    a, b = load(sys.argv[1]), load(sys.argv[2])
    calib = json.load(open(sys.argv[3]))
    ops = list(calib["clocks_per_wave_inst"].keys())
    out = {"class_of": {}, "valu_per_inst": {}, "active_quads_per_inst": {}, "clocks_per_wave_inst": calib["clocks_per_wave_inst"],
           "device": calib.get("device"), "waves_per_simd": calib.get("waves_per_simd")}
    for i, op in enumerate(ops):
        ra, rb = a.get(i, {}), b.get(i, {})
        total = ra.get("SQ_INSTS_VALU") or rb.get("SQ_INSTS_VALU")
        if not total:
            continue
        cls = "OTHER"
        for src in (ra, rb):
            for k, v in src.items():
                if k.startswith("SQ_INSTS_VALU_") and v / total >= 0.5:
                    cls = k[len("SQ_INSTS_VALU_"):]
        out["class_of"][op] = cls
        if "SQ_ACTIVE_INST_VALU" in rb:
            out["active_quads_per_inst"][op] = round(rb["SQ_ACTIVE_INST_VALU"] / rb["SQ_INSTS_VALU"], 2)
    # the union formula bench.py uses for the VALU-busy time, checked per opcode:
    # 4 x (SQ_ACTIVE_INST_VALU - SQ_ACTIVE_INST_VALU2) per instruction  vs  the measured issue clocks
    if len(sys.argv) > 5:
        c = load(sys.argv[5])
        out["busy_clocks_by_counters"] = {}
        worst = 0.0
        for i, op in enumerate(ops):
            r = c.get(i, {})
            if r.get("SQ_INSTS_VALU"):
                v = 4.0 * (r["SQ_ACTIVE_INST_VALU"] - r["SQ_ACTIVE_INST_VALU2"]) / r["SQ_INSTS_VALU"]
                out["busy_clocks_by_counters"][op] = round(v, 3)
                m = calib["clocks_per_wave_inst"][op]
                if m < 18.0:  # skip the few samples that hit a hazard stall (their issue time is not VALU time)
                    worst = max(worst, abs(v - m) / m)
        out["busy_formula_worst_relative_error"] = round(worst, 4)
        print("[opcode_classes] union formula vs measured clocks: worst relative error %.3f" % worst)
    json.dump(out, open(sys.argv[4], "w"), indent=1, sort_keys=True)
    by = collections.Counter(out["class_of"].values())
    print("[opcode_classes] %d opcodes: %s" % (len(out["class_of"]), dict(by)))


if __name__ == "__main__":
    main()
