"""Render the three views the reference's images/ directory holds (book1.png, book2.png, stanford_dragon.png) on the GPU
and save the RGB8 frames (top row first, like the PNGs) under gpurun_out/ref_views/ -- used to calibrate
tests/golden/reference_image_pins.json (tests/test_reference_image_pins.py)."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401  (first: see tests/conftest.py)

rtsr = importlib.import_module("ray-tracing-series-rust_amd")
out = os.path.join(ROOT, "gpurun_out", "ref_views")
os.makedirs(out, exist_ok=True)
VIEWS = [  # name, scene id, width, image aspect, spp, options
    ("book1_cam32", 100, 800, 1.5, 200, {"camera_aspect": 1.5}),
    ("book1_cam169", 100, 800, 1.5, 200, {}),
    ("book2", 6, 1000, 1.0, 400, {}),
    ("dragon", 11, 600, 1.6, 400, {"mesh_triangles": 200000}),
]
for name, sid, w, aspect, spp, opts in VIEWS:
    b = rtsr.Builder(1)
    world, cam, bg = b.get_world_cam(sid, **opts)
    cfg = rtsr.Config.new(aspect, w, spp, 50, 11, seed=1, background=bg)
    screen = b.flatten(world).upload().render(cam, cfg)
    np.save(os.path.join(out, name + ".npy"), screen.rgb8[::-1].copy())
    print(name, screen.rgb8.shape, screen.rgb8.mean(axis=(0, 1)), flush=True)
