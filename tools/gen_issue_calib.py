#!/usr/bin/env python3
"""Generate tools/_gen/issue_calib_auto.hip: one issue-rate micro-benchmark per VALU opcode that the SHIPPED
trace kernels contain.

    gen_issue_calib.py <device asm of render.hip (hipcc -S --cuda-device-only)> <out.hip>

For every distinct VALU mnemonic in the k_trace_* kernels one sample instruction is taken from the disassembly as
it stands (same encoding: _e32/_e64, modifiers, literals, DPP controls) and its registers are renamed onto a fixed
test file: destination -> one of 8 rotating slots (v64.., so consecutive instructions are independent), sources ->
fixed registers v96.. / s40.. that the loop never writes.  The loop body is 256 copies; tools/issue_calib_main.inc
times it with s_memtime exactly like tools/issue_calib.hip.  The result prices the kernels' instruction mix in SIMD
issue clocks (bench.py, roofline.bound = "valu").
"""
import collections
import json
import re
import sys

KERNEL_PREFIXES = ("_ZN3rtx11k_trace_", "_ZN3rtx12k_trace_", "_ZN3rtx13k_trace_", "_ZN3rtx14k_trace_", "_ZN3rtx18k_trace_")
VREG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")
SREG = re.compile(r"\bs\[(\d+):(\d+)\]|\bs(\d+)\b")


def kernel_bodies(asm_text):
    """{symbol: [instruction lines]} for every k_trace_* kernel in a device .s file."""
    out, cur, name = {}, None, None
    for line in asm_text.split("\n"):
        if cur is None:
            m = re.match(r"^(_ZN3rtx\d+k_[A-Za-z0-9_]+):", line)
            if m:
                name, cur = m.group(1), []
            continue
        if line.startswith(".Lfunc_end"):
            out[name] = cur
            cur = None
            continue
        s = line.split(";")[0].rstrip()
        if re.match(r"^\s+[a-z]", s) and not re.match(r"^\s+\.", s):
            cur.append(s.strip())
    return out


def is_valu(mn):
    return mn.startswith("v_") and not mn.startswith("v_mfma") and not mn.startswith("v_accvgpr")


def split_operands(rest):
    """Operands of an instruction, split on top-level commas; trailing modifiers (op_sel..., quad_perm:...) stay on the last."""
    ops, depth, cur = [], 0, ""
    for ch in rest:
        if ch == "[":
            depth += 1
        elif ch == "]":
            depth -= 1
        if ch == "," and depth == 0:
            ops.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        ops.append(cur.strip())
    return ops


def width_of(m):
    return (int(m.group(2)) - int(m.group(1)) + 1) if m.group(1) is not None else 1


def reg_text(prefix, base, width):
    return "%s%d" % (prefix, base) if width == 1 else "%s[%d:%d]" % (prefix, base, base + width - 1)


def rename(inst, slot):
    """Rename registers of one sample instruction for rotation slot `slot` (0..7)."""
    mn, _, rest = inst.partition(" ")
    ops = split_operands(rest)
    if not ops:
        return None
    dst = ops[0]
    dm = VREG.fullmatch(dst)
    vmap, smap = {}, {}
    next_v, next_s = [96], [40]
    if dm:
        w = width_of(dm)
        if w > 4:
            return None
        key = dm.group(0)
        vmap[key] = reg_text("v", 64 + 4 * slot, w)

    def v_sub(m):
        key = m.group(0)
        if key not in vmap:
            w = width_of(m)
            base = (next_v[0] + 1) & ~1 if w > 1 else next_v[0]
            if base + w > 128:
                raise ValueError("too many vector sources: " + inst)
            vmap[key] = reg_text("v", base, w)
            next_v[0] = base + w
        return vmap[key]

    def s_sub(m):
        key = m.group(0)
        if key not in smap:
            w = width_of(m)
            base = (next_s[0] + 1) & ~1 if w > 1 else next_s[0]
            if base + w > 60:
                raise ValueError("too many scalar sources: " + inst)
            smap[key] = reg_text("s", base, w)
            next_s[0] = base + w
        return smap[key]

    new_ops = []
    for op in ops:
        op = VREG.sub(v_sub, op)
        op = SREG.sub(s_sub, op)
        new_ops.append(op)
    return mn + " " + ", ".join(new_ops)


def main():
    asm_path, out_path = sys.argv[1], sys.argv[2]
    bodies = kernel_bodies(open(asm_path).read())
    samples = collections.OrderedDict()
    counts = collections.Counter()
    for name, body in bodies.items():
        if "k_trace" not in name:
            continue
        for inst in body:
            mn = inst.split(" ")[0]
            if not is_valu(mn):
                continue
            counts[mn] += 1
            if mn not in samples and "sdwa" not in inst and not mn.startswith("v_cmpx"):
                try:
                    if rename(inst, 0) is not None:
                        samples[mn] = inst
                except ValueError:
                    pass
    ops = sorted(samples, key=lambda k: -counts[k])
    # hand-written two-instruction bodies: what a compare + select costs through VCC and through an SGPR pair
    EXTRA = {
        "pair:v_cmp_lt_f32_e32(vcc)+v_cndmask_b32_e32(vcc)": ["v_cmp_lt_f32_e32 vcc, v96, v97", "v_cndmask_b32_e32 v{d}, v98, v99, vcc"],
        "pair:v_cmp_lt_f32_e64(sgpr)+v_cndmask_b32_e64(sgpr)": ["v_cmp_lt_f32_e64 s[40:41], v96, v97", "v_cndmask_b32_e64 v{d}, v98, v99, s[40:41]"],
        "pair:v_add_f32+v_cndmask_b32_e32(vcc)": ["v_add_f32_e32 v{d}, v96, v97", "v_cndmask_b32_e32 v{e}, v98, v99, vcc"],
        "quad:3xv_fma_f32+v_cndmask_b32_e32(vcc)": ["v_fma_f32 v{d}, v96, v97, v98", "v_fma_f32 v{e}, v96, v97, v98", "v_fma_f32 v{d}, v96, v97, v99", "v_cndmask_b32_e32 v{e}, v98, v99, vcc"],
    }
    with open(out_path, "w") as f:
        f.write("// GENERATED by tools/gen_issue_calib.py from the device assembly of the shipped kernels -- do not edit.\n")
        f.write("#include <hip/hip_runtime.h>\n")
        f.write("#define CLOB \"vcc\", \"scc\", \"memory\"" + "".join(', "v%d"' % r for r in range(64, 128)) + "".join(', "s%d"' % r for r in range(40, 60)) + "\n")
        f.write("#define N_AUTO_OPS %d\n" % (len(ops) + len(EXTRA)))
        f.write("static const char* const AUTO_OP_NAMES[] = {%s};\n" % ", ".join('"%s"' % o for o in list(ops) + list(EXTRA)))
        f.write("static const char* const AUTO_OP_SAMPLES[] = {%s};\n" % ", ".join('"%s"' % x.replace('"', "'") for x in [samples[o] for o in ops] + [" ; ".join(b) for b in EXTRA.values()]))
        f.write("template <int OP> __device__ __forceinline__ void auto_body() {\n")
        for i, mn in enumerate(ops):
            lines = [rename(samples[mn], k % 8) for k in range(256)]
            f.write("  %sif constexpr (OP == %d) {\n" % ("" if i == 0 else "else ", i))
            for c in range(0, 256, 32):
                f.write('    asm volatile("' + "\\n".join(lines[c:c + 32]) + '" ::: CLOB);\n')
            f.write("  }\n")
        base = len(ops)
        for j, (name, body) in enumerate(EXTRA.items()):
            lines = []
            for k in range(256 // len(body)):
                for inst in body:
                    lines.append(inst.format(d=64 + 4 * (k % 8), e=65 + 4 * (k % 8)))
            f.write("  else if constexpr (OP == %d) {\n" % (base + j))
            for c in range(0, 256, 32):
                f.write('    asm volatile("' + "\\n".join(lines[c:c + 32]) + '" ::: CLOB);\n')
            f.write("  }\n")
        f.write("}\n")
        f.write('#include "../issue_calib_main.inc"\n')
    json.dump({"ops": ops, "static_counts": {o: counts[o] for o in counts}}, open(out_path + ".json", "w"), indent=1)
    print("[gen_issue_calib] %d VALU opcodes (%d not benchmarkable: %s)" % (
        len(ops), len(counts) - len(ops), ", ".join(sorted(set(counts) - set(ops)))))


if __name__ == "__main__":
    main()
