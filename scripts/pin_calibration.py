"""GPU box: what the new reference-image pins measure on OUR converged frames, for a few scene seeds (the reference's own
random scene is one more draw of the same distribution): sets the tolerances of tests/test_reference_image_pins.py."""
import importlib, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401  (first: see tests/conftest.py)
rtsr = importlib.import_module("ray-tracing-series-rust_amd")
from pin_estimators import CUBE_WINDOW, sphere_cube_extents
PINS = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_image_pins.json")))
print("png cube", PINS["book2"]["sphere_cube"], "fuzzy", PINS["book2"]["regions"]["fuzzy_metal_upper"]["linear_mean"])
for sseed in (1, 2, 3):
    b = rtsr.Builder(sseed)
    world, cam, bg = b.get_world_cam(6)
    cfg = rtsr.Config.new(1.0, 1000, 1000, 50, 11, seed=sseed, background=bg, row_chunk_compat=True)
    screen = b.flatten(world).upload().render(cam, cfg)
    rad = screen.accum[::-1] / 1000.0
    r0, r1, c0, c1 = CUBE_WINDOW
    box = PINS["book2"]["regions"]["fuzzy_metal_upper"]["box"]
    clipped = np.minimum(rad, (255.5 / 256.0) ** 2)
    print("book2 scene seed", sseed, sphere_cube_extents(rad[r0:r1, c0:c1]), "fuzzy", np.round(clipped[box[0]:box[1], box[2]:box[3]].reshape(-1, 3).mean(axis=0), 4))
    # profile down a column through the cluster's bottom
    g = np.minimum(rad[..., :].min(axis=2), 0.4)
    print("   col 650 rows 490..530 min-channel:", np.round(g[490:530:3, 645:655].mean(axis=1), 3))
for sseed in (1, 2):
    b = rtsr.Builder(sseed)
    world, cam, bg = b.get_world_cam(11, mesh_triangles=200000)
    cfg = rtsr.Config.new(1.6, 600, 400, 50, 11, seed=sseed, background=bg, row_chunk_compat=True)
    screen = b.flatten(world).upload().render(cam, cfg)
    rad = np.minimum(screen.accum[::-1] / 400.0, (255.5 / 256.0) ** 2)
    for name in ("floor_far_left", "floor_far_right", "floor_near_right", "backdrop_upper"):
        reg = PINS["stanford_dragon"]["regions"][name]
        bx = reg["box"]
        got = rad[bx[0]:bx[1], bx[2]:bx[3]].reshape(-1, 3).mean(axis=0)
        print("dragon seed", sseed, name, np.round(got, 4), reg["linear_mean"], np.round(got / np.array(reg["linear_mean"]) - 1, 3))
