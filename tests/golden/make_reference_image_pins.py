"""Derive tests/golden/reference_image_pins.json from the three images the reference ships (its only outputs):

    /root/reference/images/book1.png            800 x 533   README.md:18  (canonical Book-1 final scene, an older build: gradient sky)
    /root/reference/images/book2.png           1000 x 1000  README.md:34  (Book-2 final scene, 10 000 spp, default.cfg: 11 threads)
    /root/reference/images/stanford_dragon.png  600 x 375   README.md:6   (scene 11, main.rs: Config::new(1.6, 600, ..), THREADS = 11)

Run HERE (the reference does not travel): python tests/golden/make_reference_image_pins.py
The JSON holds numbers measured on those PNGs -- silhouettes, wall boundaries, region means -- never pixels wholesale and
nothing of the reference's source.  What they can pin, although every random draw of the reference is unseeded:

  * geometry that has no randomness in it: the camera, the three big spheres of Book-1 against the sky, the dragon room's
    wall edges, the outline of Book-2's ceiling light, the rows the band split drops (world.rs:1198);
  * light transport where randomness only adds noise, as region means in LINEAR radiance (the PNG value v is
    256 * sqrt(radiance) truncated, vec3.rs:89-107; means are taken over ((v + 0.5) / 256)^2): the dragon room's walls
    (Lambertian albedo, the 4,4,4 light, the mirror ceiling, 50 bounces) far from the dragon, and Book-2's fog-lit
    background far from its random boxes and spheres.
tests/test_reference_image_pins.py compares the CPU oracle (which this pins) and the HIP path with them.
"""
import json
import os

import sys

import numpy as np
from PIL import Image

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pin_estimators import CUBE_WINDOW, sphere_cube_extents  # noqa: E402  (shared with tests/test_reference_image_pins.py)

REF = "/root/reference/images"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_image_pins.json")


def load(name):
    return np.asarray(Image.open(os.path.join(REF, name)).convert("RGB")).astype(np.int64)  # [row from the top][col][rgb]


def linear_mean(a, box):
    r0, r1, c0, c1 = box
    return [round(float(x), 6) for x in (((a[r0:r1, c0:c1] + 0.5) / 256.0) ** 2).reshape(-1, 3).mean(axis=0)]


def runs(mask_row):
    """[first, last] column of the single run of True in a row (None when empty)."""
    c = np.flatnonzero(mask_row)
    return None if len(c) == 0 else [int(c[0]), int(c[-1])]


def wall_edges(a, row):
    """Column where the left wall (G > R) gives way to the backdrop / floor (R > G), and where those give way to the right
    wall (B > R): the split that maximises the summed contrast (tests/test_reference_image_pins.py uses the same estimator)."""
    a = a.astype(np.float64)
    green_last = int(np.argmax(np.cumsum(a[row, :300, 1] - a[row, :300, 0])))
    blue_first = 300 + int(np.argmax(np.cumsum((a[row, 300:, 2] - a[row, 300:, 0])[::-1])[::-1]))
    return green_last, blue_first


def main():
    pins = {"note": "measured on /root/reference/images/*.png by tests/golden/make_reference_image_pins.py; rows count from the TOP of the image"}

    # ---- stanford_dragon.png
    d = load("stanford_dragon.png")
    edges = []
    for row in (5, 25, 50, 75, 100, 125, 150, 200, 250, 300, 350, 370):
        green_last, blue_first = wall_edges(d, row)
        edges.append({"row": row, "green_last_col": green_last, "blue_first_col": blue_first})
    regions = {"green_wall_upper": (20, 120, 5, 60), "backdrop_upper": (10, 60, 150, 450), "blue_wall_upper": (20, 120, 545, 595),
               "green_wall_lower": (250, 360, 5, 80), "blue_wall_lower": (250, 360, 520, 595),
               # the floor, XzRect y = 5, Metal (0.3, 0.3, 0.3) fuzz 0.02 (world.rs:706-713): what it mirrors of the backdrop and the
               # blue wall, lit by the (4, 4, 4) ceiling light (world.rs:739) and the mirror beside it (world.rs:714-721), far from
               # the dragon and from its reflection (the stand-in mesh is not the dragon)
               "floor_far_left": (172, 200, 100, 125), "floor_far_right": (172, 215, 445, 495), "floor_near_right": (300, 370, 410, 445)}
    black = 0
    while not d[black].any():
        black += 1
    pins["stanford_dragon"] = {
        "width": 600, "height": 375, "black_top_rows": black,
        "wall_edges": edges,
        "regions": {k: {"box": list(v), "linear_mean": linear_mean(d, v)} for k, v in regions.items()},
    }

    # ---- book2.png
    b = load("book2.png")
    black = 0
    while not b[black].any():
        black += 1
    white = b.min(axis=2) >= 250
    light_rows = {}
    for row in range(12, 146, 2):  # the ceiling light's outline: one run of saturated pixels per row
        run = runs(white[row, 90:700])
        light_rows[str(row)] = [run[0] + 90, run[1] + 90]
    regions = {"background_right_of_light": (60, 160, 720, 990), "background_right": (180, 280, 820, 990),
               "background_middle": (170, 250, 250, 480), "background_between_spheres": (420, 460, 240, 300),
               "background_left": (160, 240, 0, 45), "background_far_right": (300, 600, 940, 1000),
               # objects whose place is fixed and whose randomness averages out inside the region: the moving sphere
               # (MovingSphere, Lambertian (0.7, 0.3, 1), blurred over its 30 units of travel), the marble sphere (NoiseTexture 0.1:
               # the permutation tables are random, the mean is not), the blue subsurface sphere (glass ball + ConstantMedium 0.2)
               "moving_sphere": (290, 370, 90, 190), "marble_sphere": (430, 560, 380, 500), "subsurface_sphere": (650, 800, 200, 340),
               # the upper half of the fuzz-1.0 metal ball (Sphere (0, 150, 145) r = 50, Metal (0.8, 0.8, 0.9), world.rs:542-546;
               # hit.rs:1066-1083): reflect + 1.0 * random_in_unit_sphere of a surface that faces the ceiling light
               "fuzzy_metal_upper": (638, 678, 800, 870)}
    pins["book2"] = {
        "width": 1000, "height": 1000, "black_top_rows": black,
        "light_rows": light_rows,
        "sphere_cube": sphere_cube_extents((((b + 0.5) / 256.0) ** 2)[CUBE_WINDOW[0]:CUBE_WINDOW[1], CUBE_WINDOW[2]:CUBE_WINDOW[3]]),
        "regions": {k: {"box": list(v), "linear_mean": linear_mean(b, v)} for k, v in regions.items()},
    }

    # ---- book1.png (older build: sky = gradient white -> (0.5, 0.7, 1.0); geometry and camera as at HEAD)
    a = load("book1.png")
    nonsky = ~((a[..., 2] >= 248) & (a[..., 1] >= 222) & (a[..., 0] >= 185))
    windows = {"metal_sphere_cap": (30, 114, 480, 720), "brown_sphere_cap": (40, 114, 225, 290)}  # the horizon, with its random small spheres, begins at row ~116
    sil = {}
    for name, (r0, r1, c0, c1) in windows.items():
        rows = {}
        for row in range(r0, r1, 2):
            run = runs(nonsky[row, c0:c1])
            rows[str(row)] = None if run is None else [run[0] + c0, run[1] + c0]
        sil[name] = {"box": [r0, r1, c0, c1], "rows": rows, "count": int(nonsky[r0:r1, c0:c1].sum())}
    tops = {str(c): int(np.flatnonzero(nonsky[:, c])[0]) for c in (300, 380, 545, 600, 650)}
    pins["book1"] = {"width": 800, "height": 533, "silhouettes": sil, "first_nonsky_row_of_column": tops}

    with open(OUT, "w") as f:
        json.dump(pins, f, indent=1, sort_keys=True)
        f.write("\n")
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
