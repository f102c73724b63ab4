"""Run one C2-shaped frame under a diagnostic trace kernel (RTX_TRACE_KERNEL=vote_diag | wq_diag); counters go to stderr."""
import importlib
import os
import sys

import torch

torch.cuda.init()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rt = importlib.import_module("ray-tracing-series-rust_amd")
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 100
b = rt.Builder(scene_seed=1)
world, cam, background = b.get_world_cam(100, camera_aspect=1.5)
cfg = rt.Config.new(1.5, 800, spp, 50, 1, seed=7, background=background)
flat = b.flatten(world)
scene = flat.upload()
st = scene.render_device(cam, cfg, want_stats=True)
print("trace_ms", st.trace_ms)
