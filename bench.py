#!/usr/bin/env python3
"""Headline benchmark: Msamples/s (pixels x spp) of the path-tracing hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload c2|c1|head|c5] [--spp S]

A "step" is one full render of the workload: trace kernel(s) + ordered sample reduction + tone map,
scene already resident in HBM.  At N=1 the workload is BASELINE.json configs[1] ("c2": Book-1 final
scene, 800x533, 500 spp, depth 50).  For N>1 the SAME image is sharded by rows over the ranks (one
process per GPU, launched by torch.distributed.run) and the tone-mapped shards are gathered to rank 0
with one RCCL gather per step: total work is fixed, so scaling is "strong".

One JSON line is printed by rank 0.  Besides the driver's contract it carries
  roofline      algorithmic bytes (SURVEY.md 8d model, counts from an instrumented run of the same
                pixels/seeds) / mean trace-kernel duration measured with HIP events on the launch stream
  cpu_baseline  the CPU oracle O1 (literal restatement of the reference's path, "port") timed on this
                box's host cores on a bounded sample of the same workload
"""
import argparse
import ctypes as C
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--spp", type=int, default=0, help="override samples per pixel (marks the run as non-headline)")
    ap.add_argument("--count-spp", type=int, default=32, help="spp of the instrumented counting run (same pixels and seeds)")
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="target CPU work (wall seconds) of the CPU baseline sample")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="gloo: rehearsal on a box with fewer GPUs than ranks (shards gathered through host memory)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-count", action="store_true")
    ap.add_argument("--max-leaf", type=int, default=0, help="BVH leaf size override (0 = library default)")
    ap.add_argument("--sah-bins", type=int, default=0, help="SAH bin count override (0 = library default)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world_size != args.gpus:
        if world_size == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)" % args.gpus)
        raise SystemExit("WORLD_SIZE=%d does not match --gpus %d" % (world_size, args.gpus))
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
    sid, width, aspect, spp, depth, desc = WORKLOADS[args.workload]
    if args.spp > 0:
        spp = args.spp
    b = rtsr.Builder(1)  # scene seed 1 on every rank -> identical scene
    world, cam, bg = b.get_world_cam(sid, camera_aspect=aspect if sid in (100, 13) else 0.0)
    cfg = rtsr.Config.new(aspect, width, spp, depth, 10, seed=1, background=bg)
    height = rtsr.image_height(cfg)
    t_flat = time.perf_counter()
    flat = b.flatten(world, max_leaf=args.max_leaf, sah_bins=args.sah_bins)
    t_flat = time.perf_counter() - t_flat
    scene = flat.upload()

    shard = (rank, world_size, 1)  # rows j with j % N == rank
    my_rows = rtsr.shard_rows(cfg, shard)
    max_rows = max(rtsr.shard_rows(cfg, (r, world_size, 1)) for r in range(world_size))
    d_rgb8 = torch.zeros(max_rows * width * 3, dtype=torch.uint8, device="cuda")
    gather_list = None
    host_stage = torch.empty(d_rgb8.numel(), dtype=torch.uint8) if (world_size > 1 and args.backend == "gloo") else None
    if world_size > 1 and rank == 0:
        gather_list = [torch.empty_like(host_stage if host_stage is not None else d_rgb8) for _ in range(world_size)]
    stream = torch.cuda.current_stream().cuda_stream

    trace_ms = []
    kernel_used = []

    def step(record):
        stats = scene.render_device(cam, cfg, shard=shard, d_rgb8=d_rgb8.data_ptr(), stream=stream, want_stats=True)
        if record:
            trace_ms.append((stats.trace_ms, stats.trace_launches))
            kernel_used.append(stats.trace_kernel)
        if world_size > 1:
            if host_stage is not None:
                host_stage.copy_(d_rgb8)  # gloo rehearsal path only
                dist.gather(host_stage, gather_list=gather_list, dst=0)
            else:
                dist.gather(d_rgb8, gather_list=gather_list, dst=0)  # one RCCL gather per frame

    def fence():
        if world_size > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    fence()
    elapsed = time.perf_counter() - t0
    if world_size > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    total_samples = float(width) * height * spp  # whole job, all ranks
    value = total_samples * args.steps / elapsed / 1e6

    # ---- image assembled on rank 0 (also a sanity check of the gather)
    if rank == 0:
        img = np.zeros((height, width, 3), dtype=np.uint8)
        parts = gather_list if world_size > 1 else [d_rgb8]
        for r in range(world_size):
            rows = list(range(r, height, world_size))
            img[rows] = parts[r].cpu().numpy()[: len(rows) * width * 3].reshape(len(rows), width, 3)
        if os.environ.get("BENCH_WRITE_PPM"):
            rtsr.Screen(width, height, img).write_to_ppm_file(os.environ["BENCH_WRITE_PPM"])

    out = None
    if rank == 0:
        launches = sum(n for _, n in trace_ms)
        mean_trace_ms = sum(ms for ms, _ in trace_ms) / max(1, launches)
        roofline = None
        if not args.no_count:
            count_spp = min(spp, args.count_spp)
            ccfg = rtsr.RtxConfig.from_buffer_copy(cfg)
            ccfg.samples_per_pixel = count_spp
            st = scene.render_count(cam, ccfg, shard=shard)
            counts = {k: getattr(st, k) for k in ("box_tests", "sphere_tests", "moving_sphere_tests", "rect_tests",
                                                  "triangle_tests", "scatters", "texels", "perlin_calls", "rays", "samples")}
            bytes_per_sample = algorithmic_bytes(counts) / float(counts["samples"])
            samples_per_launch = float(my_rows) * width * spp / max(1, launches // max(1, args.steps))
            achieved = bytes_per_sample * samples_per_launch / (mean_trace_ms * 1e-3) / 1e9
            pmc = None
            valu_issue = None
            pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(pmc_path):
                try:
                    rec = json.load(open(pmc_path))
                    if rec.get("workload") == args.workload and rec.get("spp") == spp and world_size == 1:
                        pmc = rec.get("hbm_bytes_per_launch")
                        sq = rec.get("sq") or {}
                        if sq.get("SQ_INSTS_VALU") and sq.get("SQ_WAVE_CYCLES") and sq.get("SQ_WAVES"):
                            # what actually bounds this kernel (the scene is on chip): VALU issue.  A SIMD issues one
                            # wave64 VALU instruction per 4 clocks; SQ cycle counters tick once per 4 clocks.
                            simds = 4.0 * torch.cuda.get_device_properties(0).multi_processor_count
                            quad_cycles = sq["SQ_WAVE_CYCLES"] / sq["SQ_WAVES"]  # kernel duration, every wave lives through it
                            valu_issue = {"wave_insts_per_launch": sq["SQ_INSTS_VALU"],
                                          "frac_of_issue_peak": round(sq["SQ_ACTIVE_INST_VALU"] / (simds * quad_cycles), 3),
                                          "lane_utilisation": round(rec.get("lane_utilisation", 0.0), 3),
                                          "source": "profiles/pmc_traffic.json (rocprofv3 --pmc SQ_* pass of this workload)"}
                except Exception:
                    pmc = None
            roofline = {"bound": "hbm", "kernel": rtsr.trace_kernel_name(kernel_used[-1]) if kernel_used else "?", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": pmc, "valu_issue": valu_issue,
                        "bytes_per_sample": round(bytes_per_sample, 1), "kernel_ms": round(mean_trace_ms, 3),
                        "rays_per_sample": round(counts["rays"] / float(counts["samples"]), 3),
                        "box_tests_per_ray": round(counts["box_tests"] / float(max(1, counts["rays"])), 2),
                        "counted_on": "%d spp of the same pixels and seeds" % count_spp,
                        "per_ray": {k: round(v / float(max(1, counts["rays"])), 3) for k, v in counts.items() if k not in ("rays", "samples") and v}}
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
            cpu = {"value": round(width * height * cpu_spp / dt / 1e6, 4), "unit": "Msamples/s", "cores": cores, "kind": "port",
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
        out = {
            "metric": "Msamples/s (pixels x spp) on Book-1 final scene" if args.workload in ("c1", "c2", "c5") else "Msamples/s (pixels x spp)",
            "value": round(value, 2), "unit": "Msamples/s", "n_gpus": world_size, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": round(value / 1.4559, 1), "dtype": "f64", "data": "synthetic",
            "config": {"workload": desc, "width": width, "height": height, "spp": spp, "max_depth": depth,
                       "scene_seed": 1, "render_seed": 1, "flatten_s": round(t_flat, 3), "sharding": "rows j %% %d == rank, one RCCL gather of RGB8 per step" % world_size
                       if world_size > 1 else "single GPU"},
            "roofline": roofline, "cpu_baseline": cpu,
            "reference_cpu_published": {"value": 1.4559, "unit": "Msamples/s", "source": "README.md:23, 10 threads, CPU unstated"},
        }
        print(json.dumps(out), flush=True)
    if world_size > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
