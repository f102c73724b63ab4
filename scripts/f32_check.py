"""f32 fast mode against the f64 path on the catalogue scenes: image agreement and kernel time (run on the GPU box)."""
import sys, importlib, time
import numpy as np
sys.path.insert(0, "/root/repo")
rtsr = importlib.import_module("ray-tracing-series-rust_amd")

CASES = [
    ("book1", rtsr.SCENE_BOOK1_CANONICAL, 1.5, 400, 100),
    ("book1_head", rtsr.SCENE_BOOK1_HEAD, 1.5, 400, 100),
    ("cornell", rtsr.SCENE_CORNELL_BOX, 1.0, 300, 200),
    ("smoke", rtsr.SCENE_CORNELL_SMOKE, 1.0, 300, 200),
    ("book2", rtsr.SCENE_BOOK2_FINAL, 1.0, 300, 200),
    ("perlin", rtsr.SCENE_TWO_PERLIN, 1.5, 300, 100),
    ("earth", rtsr.SCENE_EARTH, 1.5, 300, 100),
    ("dragon", rtsr.SCENE_STANFORD_DRAGON, 16.0 / 9.0, 480, 64),
    ("moving", rtsr.SCENE_RANDOM_MOVING, 16.0 / 9.0, 320, 64),
]
import os
only = [a for a in sys.argv[1:] if not a.startswith("-")]
F32_ONLY = "--f32-only" in sys.argv
TINY = "--tiny" in sys.argv
for name, sid, aspect, width, spp in CASES:
    if only and name not in only:
        continue
    b = rtsr.Builder(1)
    world, cam, bg = b.get_world_cam(sid)
    flat = b.flatten(world)
    if TINY:
        width, spp = 32, 1
    cfg = rtsr.Config.new(aspect, width, spp, 50, 10, seed=1, background=bg)
    out = {}
    for mode in ((True,) if F32_ONLY else (False, True)):
        scene = flat.upload(f32=mode)
        scene.render_device(cam, cfg, want_stats=True)
        st = scene.render_device(cam, cfg, want_stats=True)
        img = scene.render(cam, cfg).accum / spp
        out[mode] = (img, st.trace_ms, rtsr.trace_kernel_name(st.trace_kernel))
        del scene
    if F32_ONLY:
        print("%-10s f32 only: kernel %s %.2f ms mean %.5f" % (name, out[True][2], out[True][1], out[True][0].mean()), flush=True)
        continue
    a, b32 = out[False][0], out[True][0]
    la, lb = a.mean(axis=2), b32.mean(axis=2)
    close = np.abs(la - lb) <= 0.02 * (np.abs(la) + 0.02)
    print("%-10s kernel %-14s f64 %8.2f ms  f32 %8.2f ms  x%.2f | mean f64 %.5f f32 %.5f rel %.2e | pixels within 2%%: %.4f | nan %d"
          % (name, out[True][2], out[False][1], out[True][1], out[False][1] / out[True][1], a.mean(), b32.mean(),
             abs(a.mean() - b32.mean()) / a.mean(), close.mean(), int(np.isnan(b32).sum())), flush=True)
