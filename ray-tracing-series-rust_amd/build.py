"""Build the HIP shared library in-tree.

    python ray-tracing-series-rust_amd/build.py            # product: lib/librtx_hip.so

hipcc cross-compiles gfx950 code objects without a GPU.  `-ffp-contract=off` is part of the
parity contract (the reference is Rust: no FMA contraction), not a tuning knob.
"""
import os
import shutil
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_DIR = os.path.dirname(PKG_DIR)
LIB_DIR = os.path.join(PKG_DIR, "lib")
LIB_PATH = os.path.join(LIB_DIR, "librtx_hip.so")
APP_PATH = os.path.join(LIB_DIR, "rtx_render")

HIP_SOURCES = [
    "csrc/hip/render.hip",
    "csrc/hip/render_f32.hip",   # render.hip again with real = float (the statistical fast mode)
    "csrc/hip/lbvh.hip",
    "csrc/hip/abi.cpp",
    "csrc/host/scene_graph.cpp",
    "csrc/host/flatten.cpp",
    "csrc/host/bvh_build.cpp",
    "csrc/host/scenes.cpp",
]
HIP_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
    "-Wall", "-Wno-unused-function",
]
# Device side of the two compilations of render.hip ONLY (the persistent trace kernels it was measured on; not the host pass, not
# lbvh.hip's radix sort, not the host sources).  MachineLICM hoists every f64 literal (two v_mov_b32) and every loop-invariant
# conversion out of the persistent loops; a 64-bit register pair built from two immediates is not rematerialisable, so the
# allocator then SPILLS those constants to scratch and reloads them inside the loops (k_trace_world: 141 dwords spilled / 264 B of
# scratch with the pass, 23 / 96 without; k_trace_lds: 125 -> 116 VGPRs).  Materialising a constant where it is used costs two
# 2-clock moves.  It is an internal LLVM option: probed once (an empty kernel), dropped if this toolchain does not know it, and
# the outcome is recorded in lib/build_flags.json (bench.py prints it).
LICM_FLAGS = ["-Xarch_device", "-mllvm=-disable-machine-licm"]
# The same two compilations, second internal option: StructurizeCFG leaves wave-UNIFORM regions alone (plain scalar branches)
# instead of rewriting them into the exec-mask form divergent regions need.  The persistent kernels are full of uniform decisions
# (the step vote, "does any lane ...", ring and queue bookkeeping), and every structurized one costs lane-mask copies and merges in
# the loops (scalar instructions are not free here).  Measured: C2 7488 -> 7621, HEAD 4688 -> 4778,
# C3 777 -> 796, C4 1076 -> 1085 Msamples/s, the whole GPU suite bit-identical.  Probed and recorded like the first.
CFG_FLAGS = ["-Xarch_device", "-mllvm=-structurizecfg-skip-uniform-regions"]
TRACE_SOURCES = ("csrc/hip/render.hip", "csrc/hip/render_f32.hip")
_probed = {}


def _accepted(flags):
    """`flags` if hipcc accepts them (compiles an empty kernel once per process), else []."""
    key = " ".join(flags)
    if key not in _probed:
        import tempfile
        with tempfile.TemporaryDirectory() as td:
            src = os.path.join(td, "probe.hip")
            with open(src, "w") as f:
                f.write("#include <hip/hip_runtime.h>\n__global__ void k(float* p) { p[threadIdx.x] = 1.f; }\n")
            rc = subprocess.run([_hipcc(), "--offload-arch=gfx950", "-O3", "-c", src, "-o", os.path.join(td, "probe.o")] + flags,
                                stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL).returncode
        _probed[key] = list(flags) if rc == 0 else []
    return _probed[key]


def licm_flags():
    return _accepted(LICM_FLAGS)


def cfg_flags():
    return _accepted(CFG_FLAGS)


def trace_flags():
    return licm_flags() + cfg_flags()


def flags_for(src):
    # RTX_EXTRA_HIPFLAGS: extra -D switches for an experiment build (scripts/ab_builds.sh); never set for the product
    return HIP_FLAGS + (trace_flags() if src in TRACE_SOURCES else []) + os.environ.get("RTX_EXTRA_HIPFLAGS", "").split()


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP library cannot be built")


def _newest(paths):
    t = 0.0
    for p in paths:
        for root, _, files in os.walk(p) if os.path.isdir(p) else [(os.path.dirname(p), [], [os.path.basename(p)])]:
            for f in files:
                fp = os.path.join(root, f)
                if os.path.exists(fp):
                    t = max(t, os.path.getmtime(fp))
    return t


def needs_build(target, deps):
    return (not os.path.exists(target)) or os.path.getmtime(target) < _newest(deps)


def build_library(force=False, verbose=True):
    deps = [os.path.join(PKG_DIR, "csrc"), os.path.join(REPO_DIR, "include")]
    if not force and not needs_build(LIB_PATH, deps):
        return LIB_PATH
    obj_dir = os.path.join(LIB_DIR, "obj")
    os.makedirs(obj_dir, exist_ok=True)
    # one object per source, compiled side by side (the two compilations of render.hip dominate), then one link
    jobs = []
    for src in HIP_SOURCES:
        obj = os.path.join(obj_dir, os.path.basename(src).rsplit(".", 1)[0] + ".o")
        if force or needs_build(obj, deps):
            cmd = [_hipcc()] + flags_for(src) + ["-c", os.path.join(PKG_DIR, src), "-o", obj]
            if verbose:
                print("[build]", " ".join(cmd), flush=True)
            jobs.append((cmd, subprocess.Popen(cmd, cwd=PKG_DIR)))
        else:
            jobs.append((None, None))
    for cmd, proc in jobs:
        if proc is not None and proc.wait() != 0:
            for _, other in jobs:
                if other is not None:
                    other.wait()
            raise subprocess.CalledProcessError(proc.returncode, cmd)
    objs = [os.path.join(obj_dir, os.path.basename(src).rsplit(".", 1)[0] + ".o") for src in HIP_SOURCES]
    cmd = [_hipcc(), "--offload-arch=gfx950", "-fPIC", "-shared"] + objs + ["-o", LIB_PATH]
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=PKG_DIR)
    import json
    with open(os.path.join(LIB_DIR, "build_flags.json"), "w") as f:
        json.dump({"hip_flags": HIP_FLAGS, "trace_kernel_flags": trace_flags(), "machine_licm_disabled": bool(licm_flags()),
                   "uniform_regions_not_structurized": bool(cfg_flags()), "trace_sources": list(TRACE_SOURCES)}, f, indent=1)
    return LIB_PATH


def build_roofline_tools(force=False, verbose=True):
    """lib/kernel_mix.json (static VALU opcode mix of every trace kernel) and lib/issue_calib (issue-cost micro-benchmark
    of exactly those opcodes), both derived from the device assembly of the sources the library was built from.
    bench.py's roofline uses them (tools/kernel_mix.py, tools/gen_issue_calib.py)."""
    mix = os.path.join(LIB_DIR, "kernel_mix.json")
    calib = os.path.join(LIB_DIR, "issue_calib")
    tools = os.path.join(REPO_DIR, "tools")
    # (tools/_gen holds this function's own intermediate files: not a dependency)
    deps = [os.path.join(PKG_DIR, "csrc")] + [os.path.join(tools, f) for f in sorted(os.listdir(tools)) if os.path.isfile(os.path.join(tools, f))]
    if not force and not needs_build(mix, deps) and not needs_build(calib, deps):
        return mix
    gen_dir = os.path.join(tools, "_gen")
    os.makedirs(gen_dir, exist_ok=True)
    asm = os.path.join(gen_dir, "render_gfx950.s")
    flags = [f for f in flags_for("csrc/hip/render.hip") if f != "-fPIC"]
    cmds = [
        [_hipcc()] + flags + ["-S", "--cuda-device-only", os.path.join(PKG_DIR, "csrc/hip/render.hip"), "-o", asm],
        [sys.executable, os.path.join(tools, "kernel_mix.py"), asm, mix],
        [sys.executable, os.path.join(tools, "gen_issue_calib.py"), asm, os.path.join(gen_dir, "issue_calib_auto.hip")],
        [_hipcc(), "--offload-arch=gfx950", "-O1", "-std=c++17", os.path.join(gen_dir, "issue_calib_auto.hip"), "-o", calib],
    ]
    for cmd in cmds:
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True, cwd=PKG_DIR)
    return mix


def build_app(force=False, verbose=True):
    """apps/rtx_render: the stand-in for the reference's src/main.rs, linked against the library."""
    src = os.path.join(PKG_DIR, "apps", "rtx_render.cpp")
    if not os.path.exists(src):
        return None
    if not force and not needs_build(APP_PATH, [src, LIB_PATH]):
        return APP_PATH
    cmd = [_hipcc(), "-O2", "-std=c++17", "-ffp-contract=off", src, "-I", os.path.join(REPO_DIR, "include"),
           "-L", LIB_DIR, "-lrtx_hip", "-Wl,-rpath,$ORIGIN", "-o", APP_PATH]
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=PKG_DIR)
    return APP_PATH


if __name__ == "__main__":
    force = "--force" in sys.argv
    build_library(force=force)
    build_app(force=force)
    build_roofline_tools(force=force)
