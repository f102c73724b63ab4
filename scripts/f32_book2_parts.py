"""Which part of the Book-2 final scene makes the f32 fast mode drift from the f64 image?  (GPU box.)"""
import sys, importlib
import numpy as np
sys.path.insert(0, "/root/repo")
rtsr = importlib.import_module("ray-tracing-series-rust_amd")

def base(b, with_boxes=True):
    objs = []
    ground = b.lambertian((0.48, 0.83, 0.53))
    if with_boxes:
        boxes = b.hittable_list()
        rng = np.random.default_rng(5)
        for i in range(20):
            for j in range(20):
                x0, z0 = -1000.0 + i * 100.0, -1000.0 + j * 100.0
                b.list_add(boxes, b.rect_prism((x0, 0.0, z0), (x0 + 100.0, float(rng.uniform(1, 101)), z0 + 100.0), ground))
        objs.append(b.bvh_from_list(boxes, 0.0, 1.0))
    objs.append(b.xz_rect(123.0, 432.0, 147.0, 412.0, 554.0, b.diffuse_light((7, 7, 7))))
    return objs

def parts(b, which):
    objs = base(b)
    if "moving" in which:
        objs.append(b.moving_sphere((400, 400, 400), (430, 400, 400), 0.0, 1.0, 50.0, b.lambertian((0.7, 0.3, 0.1))))
    if "glass" in which:
        objs.append(b.sphere((260, 150, 45), 50.0, b.dielectric(1.5)))
    if "metal" in which:
        objs.append(b.sphere((0, 150, 145), 50.0, b.metal((0.8, 0.8, 0.9), 1.0)))
    if "subsurface" in which:
        objs.append(b.sphere((360, 150, 145), 70.0, b.dielectric(1.5)))
        objs.append(b.constant_medium((0.2, 0.4, 0.9), 0.2, b.sphere((360, 150, 145), 70.0, b.dielectric(1.5))))
    if "shell" in which:
        objs.append(b.sphere((0, 0, 0), 5000.0, b.dielectric(1.5)))
    if "fog" in which:
        objs.append(b.constant_medium((1, 1, 1), 0.0001, b.sphere((0, 0, 0), 5000.0, b.dielectric(1.5))))
    if "perlin" in which:
        objs.append(b.sphere((220, 280, 300), 80.0, b.lambertian(b.noise(0.1))))
    if "cluster" in which:
        rng = np.random.default_rng(7)
        white = b.lambertian((0.73, 0.73, 0.73))
        lst = b.hittable_list([b.sphere(tuple(rng.uniform(0, 165, 3)), 10.0, white) for _ in range(1000)])
        objs.append(b.translate((-100, 270, 395), b.rotate_y(15.0, b.bvh_from_list(lst, 0.0, 1.0))))
    return b.hittable_list(objs)

CASES = [[], ["moving"], ["glass"], ["metal"], ["subsurface"], ["shell"], ["fog"], ["shell", "fog"], ["perlin"], ["cluster"]]
for which in CASES:
    b = rtsr.Builder(1)
    world = parts(b, which)
    cam = rtsr.Camera.new((478, 278, -600), (278, 278, 0), (0, 1, 0), 40.0, 1.0, 0.0, 10.0, 0.0, 1.0)
    cfg = rtsr.Config.new(1.0, 200, 200, 50, 10, seed=1, background=(0, 0, 0))
    flat = b.flatten(world)
    res = {}
    for mode in (False, True):
        sc = flat.upload(f32=mode)
        st = sc.render_device(cam, cfg, want_stats=True)
        res[mode] = (sc.render(cam, cfg).accum / 200, st.trace_ms)
    a, c = res[False][0].mean(axis=2), res[True][0].mean(axis=2)
    close = np.abs(a - c) <= 0.02 * (np.abs(a) + 0.02)
    print("%-22s f64 %7.2f ms f32 %7.2f ms | mean %.5f vs %.5f rel %+.2e | within 2%%: %.3f" %
          ("+".join(which) or "boxes+light", res[False][1], res[True][1], a.mean(), c.mean(), (c.mean() - a.mean()) / a.mean(), close.mean()), flush=True)
