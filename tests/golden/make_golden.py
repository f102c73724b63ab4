#!/usr/bin/env python3
"""Generates the golden fixtures in this directory.

What they are: per-pixel radiance sums (float64, row 0 = bottom image row) and the P3 PPM text
for a few small renders, produced by oracle O1 -- the literal CPU restatement of the reference
(oracle/o1_literal.cpp) -- at fixed scene/render seeds.  They are DATA (inputs = the parameters in
CASES, outputs = arrays), not reference source.

What they are not: outputs of the reference binary.  The reference is Rust seeded from the OS RNG and
cannot be built or run in this environment, so nothing it produced can be matched bit for bit (bit-level
parity is unpinned; what its three shipped images DO pin -- silhouettes, edges, region radiance -- is in
reference_image_pins.json, see make_reference_image_pins.py and DESIGN.md section 2).  The goldens pin THIS
build's CPU restatement so that later rounds' kernels (and the oracle itself) cannot drift silently; they
are regenerated whenever the specified random stream changes (round 2: xoroshiro128++ -> xoroshiro128+).

    python tests/golden/make_golden.py        # rewrites *.npy / *.ppm / manifest.json
"""
import hashlib
import importlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

# name: (scene id, scene seed, render seed, width, image aspect, spp, depth, scene options)
CASES = {
    "book1_canonical_64x42_4spp": (100, 1, 1, 64, 1.5, 4, 50, {}),
    "book1_head_64x36_4spp": (13, 1, 1, 64, 16 / 9, 4, 50, {}),
    "book2_final_48x48_4spp": (6, 1, 1, 48, 1.0, 4, 50, {}),
    "cornell_smoke_40x40_4spp": (5, 1, 1, 40, 1.0, 4, 50, {}),
    "triangle_test_48x27_4spp": (10, 1, 1, 48, 16 / 9, 4, 50, {}),
    "dragon_mesh8k_64x36_2spp": (11, 1, 1, 64, 16 / 9, 2, 50, {"mesh_triangles": 8000}),
}


def build_case(rtsr, name):
    sid, scene_seed, seed, width, aspect, spp, depth, opts = CASES[name]
    b = rtsr.Builder(scene_seed)
    world, cam, bg = b.get_world_cam(sid, **opts)
    cfg = rtsr.Config.new(aspect, width, spp, depth, 10, seed=seed, background=bg)
    return b, world, cam, cfg, rtsr.image_height(cfg)


def main():
    rtsr = importlib.import_module("ray-tracing-series-rust_amd")
    import oracle_py as orc
    manifest = {}
    for name in CASES:
        b, world, cam, cfg, h = build_case(rtsr, name)
        accum, rgb8 = orc.o1_render(b.graph_ptr(), world, cam, cfg, h, threads=8)
        np.save(os.path.join(HERE, name + ".accum.npy"), accum)
        ppm = os.path.join(HERE, name + ".ppm")
        rtsr.Screen(cfg.image_width, h, rgb8).write_to_ppm_file(ppm)
        manifest[name] = {"params": list(CASES[name][:7]) + [CASES[name][7]], "height": h,
                          "accum_sha256": hashlib.sha256(accum.tobytes()).hexdigest(),
                          "ppm_sha256": hashlib.sha256(open(ppm, "rb").read()).hexdigest(),
                          "mean_radiance": float(accum.mean() / cfg.samples_per_pixel)}
        print(name, manifest[name]["mean_radiance"])
    # first uniforms of a few (seed, pixel, sample) streams
    import ctypes as C
    streams = {}
    for key in [(1, 0, 0), (1, 426399, 499), (7, 123456, 3)]:
        out = np.empty(8)
        orc.load().oracle_sample_stream(key[0], key[1], key[2], 8, out.ctypes.data_as(C.POINTER(C.c_double)))
        streams["%d,%d,%d" % key] = [float(x).hex() for x in out]
    manifest["_streams"] = streams
    json.dump(manifest, open(os.path.join(HERE, "manifest.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
