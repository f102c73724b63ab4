#!/usr/bin/env python3
"""Headline benchmark: Msamples/s (pixels x spp) of the path-tracing hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload c2|c1|head|c3|c4|c5] [--spp S]

A "step" is one full render of the workload: trace kernel(s) + ordered sample reduction + tone map,
scene already resident in HBM.  Steps are pipelined two deep (two resident copies of the scene, two streams), so that
the tail of one frame overlaps the start of the next; the timed region covers K whole frames from first launch to last byte
(`single_frame` in the line = one frame at a time, what ONE rtx_render_device call delivers).  At N=1 the workload is BASELINE.json configs[1] ("c2": Book-1 final
scene, 800x533, 500 spp, depth 50).  For N>1 the SAME image is sharded by rows over the ranks (one
process per GPU, launched by torch.distributed.run) and the tone-mapped shards are gathered to rank 0
with one RCCL gather per step: total work is fixed, so scaling is "strong"; the line then also carries `c5` (one frame of
BASELINE's scaling config at a reduced, stated spp, with per-rank render and gather times) and `ranks_reported`.
`--single-process` measures the C ABI's one-process path (rtx_multi_*) instead; `--same-device` rehearses on one GPU.

One JSON line is printed by rank 0.  Besides the driver's contract it carries

  roofline      bound = "valu": the trace kernels are bound by vector-ALU ISSUE (the Book-1 scene lives in LDS; HBM
                sees 2 % of its peak).  Everything in it is measured in THIS run (rank 0, N = 1) by rocprofv3 --pmc
                passes over a child process that renders the same workload:
                  busy  = 4 x (SQ_ACTIVE_INST_VALU - SQ_ACTIVE_INST_VALU2) / (SIMDs x kernel clocks)
                        = clocks in which a SIMD's VALU is occupied by an instruction / clocks available (kernel clocks =
                          GRBM_GUI_ACTIVE / 8 XCDs of the same pass; ACTIVE_INST_VALU alone double counts quad-cycles that
                          two 2-clock instructions share -- VALU2 is exactly that overlap, see PMC_PASSES)
                  lanes = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU)
                  frac  = useful = busy x lanes          `achieved` = the matching rate of useful issue clocks
                `lane_clocks_per_sample` = busy clocks x 64 x lanes / samples: the work-based figure that falls when the
                algorithm improves.  HBM (FETCH_SIZE x 2 + WRITE_SIZE, separate passes) and the SURVEY 8(d)
                algorithmic-bytes figure ride along as secondary fields.
  cpu_baseline  the CPU oracle O1 (literal restatement of the reference's path, "port") timed on this
                box's host cores on a bounded sample of the same workload
  other_workloads  (N = 1) the other BASELINE configs on the HIP path, a few steps each: Book-1 as HEAD builds it,
                C3 (Book-2, reduced spp), C4 (871 200-triangle mesh room, full), C5's frame on one GPU (reduced spp)
"""
import argparse
import csv
import ctypes as C
import glob
import importlib
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
LIB_DIR = os.path.join(ROOT, "ray-tracing-series-rust_amd", "lib")

WORKLOADS = {
    # name: (scene id, image width, image aspect, spp, depth, description)
    "c1": (100, 200, 3.0 / 2.0, 10, 50, "Book-1 final scene, 200x133, 10 spp, depth 50"),
    "c2": (100, 800, 3.0 / 2.0, 500, 50, "Book-1 final scene, 800x533, 500 spp, depth 50"),
    "head": (13, 800, 3.0 / 2.0, 500, 50, "Book-1 scene as gen_random_scene builds it at HEAD (checker ground, moving spheres), 800x533, 500 spp"),
    "c5": (100, 3840, 16.0 / 9.0, 2000, 50, "Book-1 final scene, 3840x2160, 2000 spp, depth 50"),
    "c3": (6, 1000, 1.0, 10000, 50, "Book-2 final scene (BVH + volumes + perlin + emissives), 1000x1000, 10000 spp"),
    "c4": (11, 1920, 16.0 / 9.0, 256, 50, "dragon-class triangle mesh (871200 tris, procedural stand-in), 1920x1080, 256 spp"),
}
# fixed f64 struct sizes of the algorithmic-bytes model
S_NODE, S_SPHERE, S_MSPHERE, S_RECT, S_TRI, S_MAT, S_TEXEL, S_PERLIN, S_OUT = 64, 48, 80, 48, 112, 48, 4, 8 * 24, 24
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
N_XCD = 8              # MI355X_MICROARCH.md chip-level parameters

# rocprofv3 counter passes (<= 8 SQ counters per pass; FETCH_SIZE and WRITE_SIZE cannot share a pass)
PMC_PASSES = [
    # pass 0: VALU-active time.  SQ_ACTIVE_INST_VALU = quad-cycles (4 clocks) an instruction occupies the SIMD's VALU, counted
    # per instruction; SQ_ACTIVE_INST_VALU2 = quad-cycles in which TWO instructions are active at once (two 2-clock
    # instructions of different waves share a quad).  Their difference is the union: quad-cycles the VALU is busy.
    # Checked on the per-opcode calibration kernels (profiles/r02/classify_c.csv): 4 x (VALU - VALU2) per instruction
    # reproduces each opcode's measured issue clocks (2 / 4 / 8 / 16) within 2 %.
    ["SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VALU2", "SQ_INSTS_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAVE_CYCLES", "SQ_WAVES",
     "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "GRBM_GUI_ACTIVE", "FETCH_SIZE"],
    ["SQ_INSTS_VALU", "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_TRANS_F32",
     "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_INT64", "SQ_INSTS_VALU_CVT", "GRBM_GUI_ACTIVE", "WRITE_SIZE"],
    ["SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64",
     "SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_SALU", "GRBM_GUI_ACTIVE"],
]
VALU_CLASSES = ["ADD_F32", "MUL_F32", "FMA_F32", "TRANS_F32", "ADD_F64", "MUL_F64", "FMA_F64", "TRANS_F64", "INT32", "INT64", "CVT"]


def algorithmic_bytes(c):
    return (c["box_tests"] * S_NODE + c["sphere_tests"] * S_SPHERE + c["moving_sphere_tests"] * S_MSPHERE +
            c["rect_tests"] * S_RECT + c["triangle_tests"] * S_TRI + c["scatters"] * S_MAT + c["texels"] * S_TEXEL +
            c["perlin_calls"] * S_PERLIN + c["samples"] * S_OUT)


def usable_cores():
    """CPUs this process may use: affinity mask, capped by the cgroup CPU quota if there is one."""
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(float(quota) / float(period) + 0.5)))
    except Exception:
        pass
    return cores


# ------------------------------------------------------------------------------------------- workload set-up
def build_workload(rtsr, name, spp_override=0, max_leaf=0, sah_bins=0):
    sid, width, aspect, spp, depth, desc = WORKLOADS[name]
    if spp_override > 0:
        spp = spp_override
    b = rtsr.Builder(1)  # scene seed 1 on every rank -> identical scene
    world, cam, bg = b.get_world_cam(sid, camera_aspect=aspect if sid in (100, 13) else 0.0)
    cfg = rtsr.Config.new(aspect, width, spp, depth, 10, seed=1, background=bg)
    t0 = time.perf_counter()
    flat = b.flatten(world, max_leaf=max_leaf, sah_bins=sah_bins)
    t_flat = time.perf_counter() - t0
    return {"builder": b, "world": world, "cam": cam, "cfg": cfg, "flat": flat, "width": width, "height": rtsr.image_height(cfg),
            "spp": spp, "depth": depth, "desc": desc, "flatten_s": t_flat, "bg": bg}


# ------------------------------------------------------------------------------------------- PMC child
def pmc_child(args):
    """Runs under `rocprofv3 --pmc ...`: renders the workload a few times through the C ABI.  No torch here."""
    rtsr = importlib.import_module("ray-tracing-series-rust_amd")
    w = build_workload(rtsr, args.workload, args.spp, args.max_leaf, args.sah_bins)
    scene = w["flat"].upload()
    for _ in range(max(1, args.steps)):
        scene.render_device(w["cam"], w["cfg"], want_stats=True)  # want_stats synchronises
    return 0


def run_issue_calib(waves_per_simd=4):
    exe = os.path.join(LIB_DIR, "issue_calib")
    if not os.path.exists(exe):
        return None, "lib/issue_calib missing (python ray-tracing-series-rust_amd/build.py builds it)"
    try:
        out = subprocess.run([exe, str(waves_per_simd), "100", "3"], capture_output=True, text=True, timeout=180)
        if out.returncode != 0:
            return None, "issue_calib failed: " + out.stderr[-200:]
        return json.loads(out.stdout), None
    except Exception as e:  # noqa: BLE001
        return None, "issue_calib: %r" % (e,)


def run_pmc_passes(args, passes, keep_dir=None):
    """One rocprofv3 child per counter group; returns ({counter: mean per trace-kernel dispatch}, info)."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    info = {"passes": [], "errors": []}
    if not os.path.exists(exe):
        info["errors"].append("rocprofv3 not found")
        return {}, info
    merged, kernel_name, durations = {}, None, []
    env = dict(os.environ)
    env["TMPDIR"] = "/tmp"
    for n, counters in enumerate(passes):
        out_dir = tempfile.mkdtemp(prefix="rtx_pmc_", dir="/tmp")
        cmd = [exe, "--kernel-trace", "--pmc"] + counters + ["--output-format", "csv", "-d", out_dir, "--",
               sys.executable, os.path.join(ROOT, "bench.py"), "--pmc-child", "--workload", args.workload, "--steps", "3"]
        if args.spp > 0:
            cmd += ["--spp", str(args.spp)]
        if args.max_leaf:
            cmd += ["--max-leaf", str(args.max_leaf)]
        t0 = time.perf_counter()
        try:
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=300)
            files = glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                info["errors"].append("pass %d rc=%d: %s" % (n, r.returncode, (r.stderr or r.stdout)[-300:]))
                continue
            per_counter = {}
            for row in csv.DictReader(open(files[0])):
                if "k_trace" not in row["Kernel_Name"]:
                    continue
                kernel_name = row["Kernel_Name"].split("(")[0].replace("void ", "")
                per_counter.setdefault(row["Counter_Name"], {})[row["Dispatch_Id"]] = (
                    float(row["Counter_Value"]), (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-6)
            for cname, disp in per_counter.items():
                vals = [v for v, _ in disp.values()]
                merged.setdefault(cname, []).append(sum(vals) / len(vals))
                if cname == "GRBM_GUI_ACTIVE" or cname == counters[0]:
                    durations.append(sum(ms for _, ms in disp.values()) / len(disp))
            info["passes"].append({"counters": counters, "seconds": round(time.perf_counter() - t0, 1),
                                   "dispatches": len(next(iter(per_counter.values()))) if per_counter else 0})
            if keep_dir:
                os.makedirs(keep_dir, exist_ok=True)
                shutil.copy(files[0], os.path.join(keep_dir, "pmc_pass%d.csv" % n))
        except Exception as e:  # noqa: BLE001
            info["errors"].append("pass %d: %r" % (n, e))
        finally:
            shutil.rmtree(out_dir, ignore_errors=True)
    out = {k: sum(v) / len(v) for k, v in merged.items()}
    info["kernel"] = kernel_name
    info["pmc_kernel_ms"] = round(sum(durations) / len(durations), 3) if durations else None
    return out, info


def valu_active_fraction(pmc, pmc_info, n_cu):
    """The headline fraction: quad-cycles the SIMDs' VALUs are busy / quad-cycles available, straight from the counters."""
    need = ["SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VALU2", "GRBM_GUI_ACTIVE"]
    if any(c not in pmc for c in need):
        return None
    kernel_clocks = pmc["GRBM_GUI_ACTIVE"] / N_XCD
    simds = 4.0 * n_cu
    busy_clocks = 4.0 * (pmc["SQ_ACTIVE_INST_VALU"] - pmc["SQ_ACTIVE_INST_VALU2"])
    t_s = (pmc_info.get("pmc_kernel_ms") or 0.0) * 1e-3
    res = {"frac": round(busy_clocks / (simds * kernel_clocks), 4), "busy_clocks_per_launch": round(busy_clocks),
           "kernel_clocks": round(kernel_clocks), "simds": int(simds),
           "clock_ghz": round(kernel_clocks / t_s / 1e9, 3) if t_s > 0 else None,
           "achieved": round(busy_clocks / t_s / 1e9, 1) if t_s > 0 else None,
           "peak": round(simds * kernel_clocks / t_s / 1e9, 1) if t_s > 0 else None}
    if pmc.get("SQ_THREAD_CYCLES_VALU") and pmc.get("SQ_ACTIVE_INST_VALU"):
        # The headline fraction is USEFUL issue: `busy` counts every clock the VALU is occupied, whatever the instruction does
        # and however many of its 64 lanes are switched on; a path tracer's waves run with about half of them masked
        # (divergent walks, rejection loops), so busy alone flatters the kernel.  frac = busy x lanes on.
        res["busy"] = res["frac"]
        res["lane_utilisation"] = round(pmc["SQ_THREAD_CYCLES_VALU"] / (64.0 * pmc["SQ_ACTIVE_INST_VALU"]), 4)
        res["useful"] = round(res["busy"] * res["lane_utilisation"], 4)
        res["frac"] = res["useful"]
        if res["achieved"] is not None:
            res["achieved"] = round(res["achieved"] * res["lane_utilisation"], 1)
    if pmc.get("SQ_WAVE_CYCLES") and pmc.get("SQ_WAVES"):
        res["kernel_clocks_from_wave_cycles"] = round(4.0 * pmc["SQ_WAVE_CYCLES"] / pmc["SQ_WAVES"])
        res["wait_any_per_wave_cycle"] = round(pmc.get("SQ_WAIT_ANY", 0.0) / pmc["SQ_WAVE_CYCLES"], 4)
        res["wait_inst_any_per_wave_cycle"] = round(pmc.get("SQ_WAIT_INST_ANY", 0.0) / pmc["SQ_WAVE_CYCLES"], 4)
    return res


def build_flags():
    try:
        return json.load(open(os.path.join(LIB_DIR, "build_flags.json"))).get("trace_kernel_flags")
    except Exception:  # noqa: BLE001
        return None


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:  # noqa: BLE001
        pass
    return "unknown"


def c5_leg(rtsr, torch, dist, rank, world_size, use_gloo, spp):
    """N > 1: one frame of BASELINE's scaling config (Book-1 final scene, 3840 x 2160) at a reduced, stated spp, sharded by rows
    like the headline: per-rank render time (HIP events on the launch stream), the gather, the whole frame (max over ranks)."""
    w5 = build_workload(rtsr, "c5", spp)
    cam, cfg, width, height = w5["cam"], w5["cfg"], w5["width"], w5["height"]
    scene = w5["flat"].upload()
    shard = (rank, world_size, 1)
    max_rows = max(rtsr.shard_rows(cfg, (r, world_size, 1)) for r in range(world_size))
    buf = torch.zeros(max_rows * width * 3, dtype=torch.uint8, device="cuda")
    host = torch.empty(buf.numel(), dtype=torch.uint8) if use_gloo else None
    glist = [torch.empty_like(host if use_gloo else buf) for _ in range(world_size)] if rank == 0 else None
    stream = torch.cuda.current_stream()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]

    def frame(timed):
        if timed:
            ev[0].record(stream)
        scene.render_device(cam, cfg, shard=shard, d_rgb8=buf.data_ptr(), stream=stream.cuda_stream)
        if timed:
            ev[1].record(stream)
        if use_gloo:
            host.copy_(buf)
            dist.gather(host, gather_list=glist, dst=0)
        else:
            dist.gather(buf, gather_list=glist, dst=0)
        if timed:
            ev[2].record(stream)

    frame(False)  # warm-up: scene tables into LDS / caches, RCCL channels for this message size
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    frame(True)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    mine = torch.tensor([elapsed * 1e3, ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2])], dtype=torch.float64,
                        device="cpu" if use_gloo else "cuda")
    every = [torch.zeros_like(mine) for _ in range(world_size)]
    dist.all_gather(every, mine)
    every = [[float(x) for x in t.cpu()] for t in every]
    frame_ms = max(e[0] for e in every)
    return {"workload": w5["desc"] + " -- run at %d spp" % spp, "value": round(width * height * spp / frame_ms / 1e3, 2), "unit": "Msamples/s",
            "frame_ms": round(frame_ms, 3), "render_ms_per_rank": [round(e[1], 3) for e in every],
            "gather_ms_per_rank": [round(e[2], 3) for e in every], "gathered_bytes": world_size * max_rows * width * 3,
            "scaling": "strong (one frame cut into %d row-interleaved shards)" % world_size}


def single_process(args):
    """--single-process: N GPUs of this node from ONE process through rtx_multi_* (csrc/hip/multi.inc) -- what a Rust host that
    replaces render_scene's band threads (world.rs:1198-1244) calls.  Steps are whole frames: launches on every device, ONE
    ncclGather of the tone-mapped shards to the first device, rows back in order, copy to the host."""
    import numpy as np
    import torch  # first (see tests/conftest.py): only so that device pointers and HIP runtimes agree if anything else loads torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU render path")
    rtsr = importlib.import_module("ray-tracing-series-rust_amd")
    n = args.gpus
    n_dev = torch.cuda.device_count()
    if not args.same_device and n > n_dev:
        raise SystemExit("--single-process --gpus %d: only %d device(s) visible (add --same-device for the one-GPU rehearsal)" % (n, n_dev))
    out = {}
    for name, spp in ((args.workload, args.spp), ("c5", args.c5_spp)):
        w = build_workload(rtsr, name, spp)
        ms = rtsr.MultiScene(w["flat"], n, device_ids=[0] * n if args.same_device else list(range(n)))
        steps, warm = (args.steps, args.warmup) if name == args.workload else (1, 1)
        for _ in range(warm):
            ms.render(w["cam"], w["cfg"], want_accum=False)
        t0 = time.perf_counter()
        for _ in range(steps):
            screen = ms.render(w["cam"], w["cfg"], want_accum=False)
        elapsed = time.perf_counter() - t0
        st = screen.stats
        assert int(np.count_nonzero(screen.rgb8)) > 0
        out[name] = {"workload": w["desc"] + ("" if spp == 0 else " -- run at %d spp" % w["spp"]),
                     "value": round(float(w["width"]) * w["height"] * w["spp"] * steps / elapsed / 1e6, 2), "ms_per_step": round(elapsed / steps * 1e3, 3),
                     "render_ms_per_device": [round(st.render_ms[k], 3) for k in range(min(16, st.n_devices))], "gather_ms": round(st.gather_ms, 3),
                     "gathered_bytes": int(st.gathered_bytes), "n_shards": st.n_shards, "n_devices": st.n_devices,
                     "used_rccl": bool(st.used_rccl), "rccl_ranks_reported": st.rccl_ranks}
        del ms
    head = out[args.workload]
    line = {"metric": "Msamples/s (pixels x spp) on Book-1 final scene" if args.workload in ("c1", "c2", "c5") else "Msamples/s (pixels x spp)",
            "value": head["value"], "unit": "Msamples/s", "n_gpus": n, "steps": args.steps, "warmup": args.warmup, "ms_per_step": head["ms_per_step"],
            "higher_is_better": True, "scaling": "strong", "vs_baseline": round(head["value"] / 1.4559, 1), "dtype": "f64", "data": "synthetic",
            "config": {"workload": head["workload"], "mode": "single process, rtx_multi_* (C ABI)" + (", all shards on device 0 (rehearsal, no collective)" if args.same_device else ""),
                       "frames_in_flight": 1, "sharding": "rows j %% %d == shard, one ncclGather of RGB8 per frame, host copy included" % n},
            "roofline": None, "cpu_baseline": None, "multi": head, "c5": out.get("c5") if args.workload != "c5" else None}
    print(json.dumps(line), flush=True)
    return 0


# ------------------------------------------------------------------------------------------- main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--spp", type=int, default=0, help="override samples per pixel (marks the run as non-headline)")
    ap.add_argument("--count-spp", type=int, default=32, help="spp of the instrumented counting run (same pixels and seeds)")
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="target CPU work (wall seconds) of the CPU baseline sample")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="gloo: rehearsal on a box with fewer GPUs than ranks (shards gathered through host memory)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-count", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 counter passes (roofline then reports no VALU fraction)")
    ap.add_argument("--keep-pmc", default="", help="directory to keep the raw counter CSVs in")
    ap.add_argument("--no-extras", action="store_true", help="skip the other BASELINE configs")
    ap.add_argument("--shard-of", type=int, default=0, help="analysis only (N = 1): render just rank 0's shard of an N-GPU job per step; "
                    "`value` is then N x this GPU's rate = what N GPUs would deliver before the gather")
    ap.add_argument("--frames-in-flight", type=int, default=2, choices=[1, 2], help="frames pipelined (2: the tail, reduction, tone map and "
                    "gather of frame k overlap the start of frame k + 1; 1: one frame at a time, e.g. under rocprofv3 --kernel-trace, whose "
                    "start stamp of a queued kernel is taken before its waves can run)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--single-process", action="store_true", help="N GPUs from ONE process through the C ABI's rtx_multi_* (one stream per "
                    "device, ncclCommInitAll, one ncclGather per frame): the path a Rust host calls.  Launch WITHOUT torch.distributed.run")
    ap.add_argument("--same-device", action="store_true", help="with --single-process: all N shards on device 0, copies instead of the "
                    "collective (rehearsal of the sharding on a one-GPU box)")
    ap.add_argument("--c5-spp", type=int, default=96, help="N > 1: spp of the extra C5 frame (3840x2160; BASELINE's scaling config at reduced spp)")
    ap.add_argument("--max-leaf", type=int, default=0, help="BVH leaf size override (0 = library default)")
    ap.add_argument("--sah-bins", type=int, default=0, help="SAH bin count override (0 = library default)")
    args = ap.parse_args()
    if args.pmc_child:
        return pmc_child(args)

    if args.single_process:
        return single_process(args)

    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world_size != args.gpus:
        if world_size == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)" % args.gpus)
        raise SystemExit("WORLD_SIZE=%d does not match --gpus %d" % (world_size, args.gpus))

    # ---- measurements that run in child processes come first: this process has not touched the GPU yet
    pmc, pmc_info = {}, {"errors": ["skipped"]}
    if rank == 0 and world_size == 1 and not args.no_pmc:
        pmc, pmc_info = run_pmc_passes(args, PMC_PASSES, keep_dir=args.keep_pmc or None)
        if args.keep_pmc:  # the per-opcode issue costs behind DESIGN.md 5.1's check of the formula (lib/issue_calib)
            calib, _ = run_issue_calib(4)
            if calib is not None:
                json.dump(calib, open(os.path.join(args.keep_pmc, "issue_calib.json"), "w"))

    import numpy as np
    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU render path")
    n_dev = torch.cuda.device_count()
    device_index = local_rank if args.backend == "nccl" else local_rank % max(1, n_dev)
    torch.cuda.set_device(device_index)
    if world_size > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world_size,
                                    device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world_size)

    rtsr = importlib.import_module("ray-tracing-series-rust_amd")
    w = build_workload(rtsr, args.workload, args.spp, args.max_leaf, args.sah_bins)
    b, world, cam, cfg, flat = w["builder"], w["world"], w["cam"], w["cfg"], w["flat"]
    width, height, spp, depth, desc = w["width"], w["height"], w["spp"], w["depth"], w["desc"]
    n_cu = torch.cuda.get_device_properties(device_index).multi_processor_count

    shard = (rank, world_size, 1)  # rows j with j % N == rank
    emulated = args.shard_of if (world_size == 1 and args.shard_of > 1) else 0
    if emulated:
        shard = (0, emulated, 1)
    my_rows = rtsr.shard_rows(cfg, shard)
    max_rows = max(rtsr.shard_rows(cfg, (r, shard[1], 1)) for r in range(shard[1]))
    # Frames are pipelined two deep: consecutive steps alternate between two resident copies of the scene (each owns its
    # render workspace), two streams and two output buffers, so the tail of frame k -- its few 50-bounce paths in otherwise idle
    # waves, the ordered reduction, the tone map and (N > 1) the gather -- overlaps the start of frame k + 1.  Nothing is skipped:
    # every step still traces, reduces, tone-maps and gathers one full frame; the timed region ends when all of them have finished.
    N_PIPE = args.frames_in_flight
    scenes = [flat.upload() for _ in range(N_PIPE)]
    scene = scenes[0]
    streams = [torch.cuda.Stream() for _ in range(N_PIPE)]
    bufs = [torch.zeros(max_rows * width * 3, dtype=torch.uint8, device="cuda") for _ in range(N_PIPE)]
    d_rgb8 = bufs[0]
    use_gloo = world_size > 1 and args.backend == "gloo"
    host_stage = [torch.empty(bufs[0].numel(), dtype=torch.uint8) for _ in range(N_PIPE)] if use_gloo else None
    gather_lists = [None] * N_PIPE
    if world_size > 1 and rank == 0:
        gather_lists = [[torch.empty_like(host_stage[0] if use_gloo else bufs[0]) for _ in range(world_size)] for _ in range(N_PIPE)]
    pending = [None] * N_PIPE
    step_no = [0]

    def step():
        p = step_no[0] % N_PIPE
        step_no[0] += 1
        with torch.cuda.stream(streams[p]):
            if pending[p] is not None:
                pending[p].wait()  # the gather that last read this buffer (stream-side wait, the host goes on)
                pending[p] = None
            scenes[p].render_device(cam, cfg, shard=shard, d_rgb8=bufs[p].data_ptr(), stream=streams[p].cuda_stream)
            if world_size > 1:
                if use_gloo:
                    host_stage[p].copy_(bufs[p])  # gloo rehearsal path only (synchronises)
                    dist.gather(host_stage[p], gather_list=gather_lists[p], dst=0)
                else:
                    pending[p] = dist.gather(bufs[p], gather_list=gather_lists[p], dst=0, async_op=True)  # one RCCL gather per frame

    def fence():
        for p in range(N_PIPE):
            if pending[p] is not None:
                with torch.cuda.stream(streams[p]):
                    pending[p].wait()
                pending[p] = None
        if world_size > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if world_size > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    last = (step_no[0] - 1) % N_PIPE
    d_rgb8 = bufs[last]
    gather_list = gather_lists[last]
    # the trace kernel's own duration: HIP events around its launches on the launch stream, in three untimed frames of the same
    # shard run one at a time (inside the pipelined region the events of two streams would interleave)
    trace_ms = []
    kernel_used = []
    stream = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        stats = scene.render_device(cam, cfg, shard=shard, d_rgb8=bufs[0].data_ptr(), stream=stream, want_stats=True)
        trace_ms.append((stats.trace_ms, stats.trace_launches))
        kernel_used.append(stats.trace_kernel)
    torch.cuda.synchronize()

    total_samples = float(width) * height * spp  # whole job, all ranks
    value = total_samples * args.steps / elapsed / 1e6
    # one frame alone (no pipelining): what ONE rtx_render call delivers -- the reference's metric is one render
    single_ms = None
    if world_size == 1:
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(3):
            scene.render_device(cam, cfg, shard=shard, d_rgb8=bufs[0].data_ptr(), stream=stream)
            torch.cuda.synchronize()
        single_ms = (time.perf_counter() - t1) / 3 * 1e3
    c5 = None
    if world_size > 1:
        try:
            c5 = c5_leg(rtsr, torch, dist, rank, world_size, use_gloo, args.c5_spp)
        except Exception as e:  # noqa: BLE001  (the headline line must still be printed)
            c5 = {"error": repr(e)[:300]}

    # ---- image assembled on rank 0 (also a sanity check of the gather)
    if rank == 0:
        img = np.zeros((height, width, 3), dtype=np.uint8)
        parts = gather_list if world_size > 1 else [d_rgb8]
        for r in range(world_size):
            rows = list(range(r, height, shard[1]))
            img[rows] = parts[r].cpu().numpy()[: len(rows) * width * 3].reshape(len(rows), width, 3)
        if os.environ.get("BENCH_WRITE_PPM"):
            rtsr.Screen(width, height, img).write_to_ppm_file(os.environ["BENCH_WRITE_PPM"])

    out = None
    if rank == 0:
        launches = sum(n for _, n in trace_ms)
        mean_trace_ms = sum(ms for ms, _ in trace_ms) / max(1, launches)
        kernel_name = rtsr.trace_kernel_name(kernel_used[-1]) if kernel_used else "?"
        roofline = {"bound": "valu", "kernel": kernel_name, "achieved": None, "peak": None, "unit": "G issue-clk/s (VALU issue clocks over all SIMDs)",
                    "frac": None, "traffic": None, "kernel_ms": round(mean_trace_ms, 3)}
        # -- VALU issue roofline from this run's counter passes + calibration
        if pmc:
            va = valu_active_fraction(pmc, pmc_info, n_cu)
            if va is not None:
                roofline.update(va)
                roofline["kernel"] = pmc_info.get("kernel") or kernel_name
                roofline["pmc_kernel_ms"] = pmc_info.get("pmc_kernel_ms")
                roofline["formula"] = ("busy = 4 x (SQ_ACTIVE_INST_VALU - SQ_ACTIVE_INST_VALU2) / (SIMDs x GRBM_GUI_ACTIVE / 8); "
                                       "lane_utilisation = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU); frac = useful = busy x lane_utilisation")
                if va.get("lane_utilisation"):
                    # a work-based figure that must FALL when the algorithm gets better (frac alone cannot tell less work from idling):
                    # lane-clocks of VALU issue spent per camera path
                    roofline["lane_clocks_per_sample"] = round(va["busy_clocks_per_launch"] * 64.0 * va["lane_utilisation"] / (float(my_rows) * width * spp), 1)
                roofline["source"] = ("measured in this run: %d rocprofv3 --pmc passes over a child process rendering the same workload" %
                                      len(pmc_info.get("passes", [])))
            if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
                # MI355X_MICROARCH.md, HBM: FETCH_SIZE (KB) under-reports wide reads 2x on gfx950; WRITE_SIZE (KB) is exact
                hbm_bytes = pmc["FETCH_SIZE"] * 1024.0 * 2.0 + pmc["WRITE_SIZE"] * 1024.0
                roofline["traffic"] = round(hbm_bytes)
                t_s = (pmc_info.get("pmc_kernel_ms") or mean_trace_ms) * 1e-3
                roofline["hbm"] = {"achieved": round(hbm_bytes / t_s / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": round(hbm_bytes / t_s / 1e9 / HBM_PEAK_GBS, 4)}
            if pmc.get("SQ_ACTIVE_INST_LDS"):
                roofline["lds"] = {"bank_conflict_per_active_cycle": round(pmc.get("SQ_LDS_BANK_CONFLICT", 0.0) / pmc["SQ_ACTIVE_INST_LDS"], 3),
                                   "insts_per_launch": round(pmc.get("SQ_INSTS_LDS", 0.0))}
        if pmc_info.get("errors") and pmc_info["errors"] != ["skipped"]:
            roofline["pmc_errors"] = pmc_info["errors"][:3]
        # -- SURVEY 8(d) algorithmic bytes (a model figure: the scene is on chip, so it exceeds what HBM could deliver)
        if not args.no_count:
            count_spp = min(spp, args.count_spp)
            ccfg = rtsr.RtxConfig.from_buffer_copy(cfg)
            ccfg.samples_per_pixel = count_spp
            st = scene.render_count(cam, ccfg, shard=shard)
            counts = {k: getattr(st, k) for k in ("box_tests", "sphere_tests", "moving_sphere_tests", "rect_tests",
                                                  "triangle_tests", "scatters", "texels", "perlin_calls", "rays", "samples")}
            bytes_per_sample = algorithmic_bytes(counts) / float(counts["samples"])
            samples_per_launch = float(my_rows) * width * spp / max(1, launches // max(1, len(trace_ms)))
            achieved = bytes_per_sample * samples_per_launch / (mean_trace_ms * 1e-3) / 1e9
            roofline["algorithmic"] = {"bytes_per_sample": round(bytes_per_sample, 1), "achieved": round(achieved, 1), "unit": "GB/s",
                                       "over_hbm_peak": round(achieved / HBM_PEAK_GBS, 3),
                                       "note": "SURVEY 8(d) model (every touch as if from memory); > 1 of HBM peak because the scene is LDS/L2 resident",
                                       "rays_per_sample": round(counts["rays"] / float(counts["samples"]), 3),
                                       "box_tests_per_ray": round(counts["box_tests"] / float(max(1, counts["rays"])), 2),
                                       "counted_on": "%d spp of the same pixels and seeds" % count_spp}
        # -- the other BASELINE configs on the HIP path (short runs; C2 stays the headline)
        extras = None
        if world_size == 1 and not args.no_extras and args.workload == "c2" and args.spp == 0:
            extras = {}
            scenes.clear()
            del scene  # free C2's sample buffers first
            torch.cuda.empty_cache()
            # "<name>_f32": the same workload through the statistical fast mode (rtx_scene_upload_f32) -- reported beside the
            # f64 numbers, never instead of them: `value` and `dtype` of this line are the bit-exact f64 path
            built = {}
            for name, x_spp, x_steps in (("c2_f32", 0, 3), ("head", 0, 3), ("head_f32", 0, 3), ("c3", 256, 2), ("c3_f32", 256, 2),
                                         ("c4", 0, 2), ("c4_f32", 0, 2), ("c5", 120, 1)):
                try:
                    f32 = name.endswith("_f32")
                    wl = name[:-4] if f32 else name
                    if wl not in built:
                        built.clear()  # one flattened scene alive at a time (the dragon room is 0.3 GB of host arrays)
                        built[wl] = build_workload(rtsr, wl, x_spp)
                    xw = built[wl]
                    t_up = time.perf_counter()
                    xs = xw["flat"].upload(f32=f32)
                    t_up = time.perf_counter() - t_up
                    xs.render_device(xw["cam"], xw["cfg"], stream=stream, want_stats=True)  # warm-up
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    ms, kern = 0.0, None
                    for _ in range(x_steps):
                        st = xs.render_device(xw["cam"], xw["cfg"], stream=stream, want_stats=True)
                        ms += st.trace_ms
                        kern = st.trace_kernel
                    torch.cuda.synchronize()
                    dt = (time.perf_counter() - t1) / x_steps
                    extras[name] = {"workload": xw["desc"] + ("" if x_spp == 0 else " -- run at %d spp" % x_spp),
                                    "value": round(xw["width"] * xw["height"] * xw["spp"] / dt / 1e6, 1), "unit": "Msamples/s",
                                    "ms_per_step": round(dt * 1e3, 2), "trace_ms": round(ms / x_steps, 2), "kernel": rtsr.trace_kernel_name(kern),
                                    "flatten_s": round(xw["flatten_s"], 2), "upload_s": round(t_up, 2), "dtype": "f32 (statistical fast mode)" if f32 else "f64"}
                    del xs
                except Exception as e:  # noqa: BLE001
                    extras[name] = {"error": repr(e)[:200]}
        cpu = None
        if not args.no_cpu_baseline and world_size == 1:  # a reported baseline: rank 0 at N=1 only
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import oracle_py as orc
            ccfg = rtsr.RtxConfig.from_buffer_copy(cfg)
            cores = usable_cores()
            # calibrate on 1 spp, then size the sample to ~args.cpu_seconds of wall time (bounded by the full spp)
            ccfg.samples_per_pixel = 1
            t0 = time.perf_counter()
            orc.o1_render(b.graph_ptr(), world, cam, ccfg, height, threads=cores)
            dt1 = max(1e-3, time.perf_counter() - t0)
            cpu_spp = int(max(1, min(spp, round(args.cpu_seconds / dt1))))
            ccfg.samples_per_pixel = cpu_spp
            t0 = time.perf_counter()
            orc.o1_render(b.graph_ptr(), world, cam, ccfg, height, threads=cores)
            dt = time.perf_counter() - t0
            cpu = {"value": round(width * height * cpu_spp / dt / 1e6, 4), "unit": "Msamples/s", "cores": cores, "kind": "port", "cpu_model": cpu_model(),
                   "sample": "%dx%d at %d spp (same scene, camera, depth, seeds), oracle O1 with %d row-band threads, %.1f s" % (
                       width, height, cpu_spp, cores, dt)}
            # the README's setting (10 threads, README.md:23), on a smaller sample of the same workload
            spp10 = int(max(1, min(spp, round(cpu_spp * 10.0 / max(10, cores) * 0.5))))
            ccfg.samples_per_pixel = spp10
            t0 = time.perf_counter()
            orc.o1_render(b.graph_ptr(), world, cam, ccfg, height, threads=10)
            dt10 = time.perf_counter() - t0
            cpu["ten_threads"] = {"value": round(width * height * spp10 / dt10 / 1e6, 4), "cores": 10,
                                  "sample": "%d spp, %.1f s" % (spp10, dt10)}
            cpu["gpu_over_cpu"] = {"vs_all_cores": round(value / cpu["value"], 1), "vs_ten_threads": round(value / cpu["ten_threads"]["value"], 1),
                                   "vs_readme_ten_threads": round(value / 1.4559, 1)}
        out = {
            "metric": "Msamples/s (pixels x spp) on Book-1 final scene" if args.workload in ("c1", "c2", "c5") else "Msamples/s (pixels x spp)",
            "value": round(value, 2), "unit": "Msamples/s", "n_gpus": world_size, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": round(value / 1.4559, 1), "dtype": "f64", "data": "synthetic",
            "config": {"workload": desc, "width": width, "height": height, "spp": spp, "max_depth": depth,
                       "scene_seed": 1, "render_seed": 1, "flatten_s": round(w["flatten_s"], 3), "frames_in_flight": N_PIPE, "sharding": "rows j %% %d == rank, one RCCL gather of RGB8 per step" % world_size
                       if world_size > 1 else "single GPU"},
            "roofline": roofline, "cpu_baseline": cpu,
            "reference_cpu_published": {"value": 1.4559, "unit": "Msamples/s", "source": "README.md:23, 10 threads, CPU unstated"},
        }
        if single_ms is not None:
            out["single_frame"] = {"ms": round(single_ms, 3), "value": round(total_samples / single_ms / 1e3, 2), "unit": "Msamples/s",
                                   "note": "one frame at a time, launch to last byte (what one rtx_render_device call takes); `value` above pipelines %d frames" % N_PIPE}
        if world_size > 1:
            out["ranks_reported"] = {"torch.distributed": dist.get_world_size(), "backend": dist.get_backend()}
            out["c5"] = c5
        out["build_flags"] = build_flags()
        if extras is not None:
            out["other_workloads"] = extras
        if emulated:
            out["emulated_ranks"] = emulated
            out["note"] = "analysis run: one GPU rendered rank 0's shard of a %d-GPU job; value = %d x its rate, no gather" % (emulated, emulated)
        print(json.dumps(out), flush=True)
    if world_size > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
