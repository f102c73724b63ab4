"""Per-GPU step time of the C2 frame when it is cut into N row shards (what each rank of an N-GPU run does),
measured on one GPU: python scripts/shard_time.py 1 2 4 8"""
import importlib
import os
import sys
import time

import torch

torch.cuda.init()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rt = importlib.import_module("ray-tracing-series-rust_amd")
b = rt.Builder(1)
world, cam, bg = b.get_world_cam(100, camera_aspect=1.5)
cfg = rt.Config.new(1.5, 800, 500, 50, 10, seed=1, background=bg)
scene = b.flatten(world).upload()
h = rt.image_height(cfg)
for n in [int(a) for a in sys.argv[1:]] or [1, 8]:
    rows = rt.shard_rows(cfg, (0, n, 1))
    d = torch.zeros(rows * 800 * 3, dtype=torch.uint8, device="cuda")
    best = 1e9
    for _ in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        st = scene.render_device(cam, cfg, shard=(0, n, 1), d_rgb8=d.data_ptr(), stream=torch.cuda.current_stream().cuda_stream, want_stats=True)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    print("N=%d: %d rows, step %.3f ms (trace kernel %.3f ms); ideal %.3f ms; efficiency %.1f %%" % (
        n, rows, best * 1e3, st.trace_ms, 48.6 / n, 100 * (48.6 / n) / (best * 1e3)))
