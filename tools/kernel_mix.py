#!/usr/bin/env python3
"""Static VALU opcode histogram of every trace kernel in the shipped code object.

    kernel_mix.py <device asm of render.hip> <out.json>

Output: {"kernels": {demangled name up to "(": {"symbol": ..., "valu": {mnemonic: static count}, "n_valu": N,
"n_inst": M, "vgpr": .., "sgpr_spill": .., "vgpr_spill": .., "scratch": ..}}}.  bench.py weights the measured per-opcode issue
costs with these counts inside each hardware counter class (roofline.bound = "valu").
"""
import collections
import json
import re
import subprocess
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from gen_issue_calib import is_valu, kernel_bodies  # noqa: E402


def demangle(sym):
    for tool in ("/opt/rocm/lib/llvm/bin/llvm-cxxfilt", "c++filt"):
        try:
            return subprocess.run([tool, sym], capture_output=True, text=True, check=True).stdout.strip()
        except Exception:
            continue
    return sym


def main():
    text = open(sys.argv[1]).read()
    bodies = kernel_bodies(text)
    # resource usage comments the compiler leaves after each kernel
    meta = {}
    for m in re.finditer(r"^\s*\.amdhsa_kernel (\S+)\n(.*?)\.end_amdhsa_kernel", text, re.S | re.M):
        body = m.group(2)
        def field(name, default=0):
            mm = re.search(r"\.amdhsa_%s (\S+)" % name, body)
            try:
                return int(mm.group(1), 0) if mm else default
            except ValueError:
                return default
        meta[m.group(1)] = {"next_free_vgpr": field("next_free_vgpr"), "next_free_sgpr": field("next_free_sgpr"),
                            "scratch": field("private_segment_fixed_size"), "lds_static": field("group_segment_fixed_size")}
    spills = {}
    for m in re.finditer(r"; Function info:.*?\n(.*?)(?=\n\t\.|\n_Z|\Z)", text, re.S):
        pass
    for sym in bodies:
        mm = re.search(re.escape(sym) + r":.*?; sgpr spill count: (\d+).*?; vgpr spill count: (\d+)", text, re.S)
    out = {}
    for sym, body in bodies.items():
        if "k_trace" not in sym:
            continue
        hist = collections.Counter(i.split(" ")[0] for i in body if is_valu(i.split(" ")[0]))
        name = demangle(sym).split("(")[0].replace("void ", "")
        rec = {"symbol": sym, "valu": dict(hist), "n_valu": sum(hist.values()), "n_inst": len(body)}
        rec.update(meta.get(sym, {}))
        # "; ScratchSize: N" / "; NumVgprs: N" / spill counts follow the kernel body as comments
        tail = text.split(sym + ":", 1)[1].split(".Lfunc_end", 1)[1][:8000] if sym + ":" in text else ""
        for key, pat in (("vgprs", r"; NumVgprs: (\d+)"), ("sgprs", r"; NumSgprs: (\d+)"), ("scratch_size", r"; ScratchSize: (\d+)"),
                         ("occupancy", r"; Occupancy: (\d+)"), ("sgpr_spill", r"; SGPRSpillCount: (\d+)|; sgpr spill count: (\d+)"),
                         ("vgpr_spill", r"; VGPRSpillCount: (\d+)|; vgpr spill count: (\d+)")):
            mm = re.search(pat, tail)
            if mm:
                rec[key] = int([g for g in mm.groups() if g is not None][0])
        out[name] = rec
    json.dump({"kernels": out}, open(sys.argv[2], "w"), indent=1, sort_keys=True)
    print("[kernel_mix] %d kernels" % len(out))


if __name__ == "__main__":
    main()
