"""Bit-exact A/B of trace-kernel variants on the GPU box (the RTX_* environment is read at scene upload).

    python scripts/kernel_parity.py RTX_TRACE_KERNEL=wq [time]    # variant vs baseline, then timed C2 frames
    python scripts/kernel_parity.py RTX_RING=1 time

The baseline is the plain voting kernel (RTX_RING=0 RTX_SCENE_LDS=0, no RTX_TRACE_KERNEL), which tests/test_gpu_parity.py pins
against the CPU oracle.
"""
import importlib
import os
import sys
import time

import numpy as np
import torch

torch.cuda.init()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rt = importlib.import_module("ray-tracing-series-rust_amd")


def render(kernel, scene_id, aspect, width, spp, depth, seed=7, timed=0):
    for k in [k for k in os.environ if k.startswith("RTX_")]:
        os.environ.pop(k)
    if kernel:
        for kv in kernel.split(","):
            k, v = kv.split("=")
            os.environ[k] = v
    else:
        os.environ["RTX_RING"] = "0"
        os.environ["RTX_SCENE_LDS"] = "0"
    b = rt.Builder(scene_seed=1)
    world, cam, background = b.get_world_cam(scene_id, camera_aspect=aspect)
    cfg = rt.Config.new(aspect, width, spp, depth, 1, seed=seed, background=background)
    flat = b.flatten(world)
    scene = flat.upload()
    screen = scene.render(cam, cfg, want_accum=True)
    best = None
    for _ in range(timed):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        st = scene.render_device(cam, cfg, want_stats=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ms = st.trace_ms if hasattr(st, "trace_ms") else dt * 1e3
        best = ms if best is None else min(best, ms)
    return np.array(screen.accum, copy=True), best


if __name__ == "__main__":
    variant = sys.argv[1] if len(sys.argv) > 1 else "RTX_RING=1"
    ok = True
    for (sid, aspect, w, spp, depth) in [(100, 1.5, 96, 8, 50), (100, 1.5, 300, 16, 50), (13, 16 / 9, 200, 16, 50), (7, 16 / 9, 200, 8, 50), (9, 16 / 9, 200, 8, 50), (100, 1.5, 64, 4, 1)]:
        a, _ = render(None, sid, aspect, w, spp, depth)
        v, _ = render(variant, sid, aspect, w, spp, depth)
        same = a.tobytes() == v.tobytes()
        ok &= same
        print(f"scene {sid} {w}px {spp}spp: {variant} == baseline: {same}  (max |diff| {np.abs(a - v).max():.3g})", flush=True)
    if not ok:
        sys.exit(1)
    if len(sys.argv) > 2 and sys.argv[2] == "time":
        for k in (None, variant):
            _, ms = render(k, 100, 1.5, 800, 500, 50, timed=3)
            print(f"C2 800x533x500: kernel={k or 'baseline'} trace_ms={ms:.2f}  => {800*533*500/ms/1e3:.0f} Msamples/s", flush=True)
