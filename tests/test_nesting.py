"""Containers inside containers: what the Hittable trait allows (hit.rs:82-85, bvh.rs:85-93) and the flat scene has to express.

  * a BvhNode whose list holds lists and other BvhNodes   -> one tree over all the primitives below
  * Translate / RotateY chains of up to RT_MAX_XFORM_OPS = 4 wrappers around one object (hit.rs:787-936)
  * lists nested in the world list, media anywhere in them (they dissolve into the world list in order)
Each world is rendered by the literal object-graph oracle O1 (which recurses through the containers as the reference does), by
the flat-array oracle O2 and -- on the GPU box -- by the HIP path: bit-identical.  What stays unsupported is reported
(tests/test_abi_and_errors.py): a Translate / RotateY / ConstantMedium INSIDE a BVH, five or more wrappers, a medium of a medium.
"""
import numpy as np
import pytest


def nested_worlds(rtsr):
    def base(seed=1):
        b = rtsr.Builder(seed)
        mats = [b.lambertian((0.8, 0.3, 0.3)), b.metal((0.8, 0.8, 0.9), 0.3), b.dielectric(1.5), b.lambertian((0.2, 0.7, 0.3)),
                b.diffuse_light((3.0, 3.0, 3.0))]
        return b, mats

    def balls(b, mats, n, x0, z0, r=0.35):
        out = []
        for k in range(n):
            out.append(b.sphere((x0 + 0.9 * (k % 5) + 0.13 * b.random(), 0.4 + 0.8 * (k // 10) + 0.1 * b.random(), z0 + 0.9 * ((k // 5) % 2) + 0.2 * b.random()),
                                r * (0.7 + 0.5 * b.random()), mats[k % 4]))
        return out

    # 1: BVH( sphere, list(spheres), BVH(spheres), BVH(list(BVH(spheres), rect prism)) ) + a ground sphere in the world list
    b, m = base()
    inner1 = b.bvh_from_list(b.hittable_list(balls(b, m, 12, -3.0, -1.0)), 0.0, 1.0)
    inner2 = b.bvh_from_list(b.hittable_list(balls(b, m, 9, -1.0, 1.2)), 0.0, 1.0)
    mixed = b.bvh_from_list(b.hittable_list([inner2, b.rect_prism((2.2, 0.0, -0.5), (3.0, 1.3, 0.4), m[3])]), 0.0, 1.0)
    outer = b.bvh_from_list(b.hittable_list([b.sphere((0.0, 2.6, 0.0), 0.6, m[4]), b.hittable_list(balls(b, m, 7, 0.5, -2.0)), inner1, mixed]), 0.0, 1.0)
    yield "bvh_in_bvh", b, b.hittable_list([b.sphere((0.0, -500.0, 0.0), 500.0, m[0]), outer])

    # 2: chains of three and four wrappers around a box, a sphere BVH and a list
    b, m = base(2)
    box = b.rect_prism((0.0, 0.0, 0.0), (1.0, 1.6, 1.0), m[1])
    chain3 = b.translate((-2.5, 0.0, 0.5), b.rotate_y(25.0, b.translate((0.2, 0.1, -0.3), box)))
    cluster = b.bvh_from_list(b.hittable_list(balls(b, m, 10, 0.0, 0.0, 0.25)), 0.0, 1.0)
    chain4 = b.rotate_y(-20.0, b.translate((1.0, 0.2, -1.0), b.rotate_y(35.0, b.translate((-0.5, 0.0, 0.4), cluster))))
    chain4_list = b.translate((0.3, 0.0, 2.0), b.translate((0.2, 0.3, 0.1), b.rotate_y(10.0, b.rotate_y(-55.0, b.hittable_list(balls(b, m, 4, -1.0, 0.0))))))
    yield "xform_chains", b, b.hittable_list([b.sphere((0.0, -500.0, 0.0), 500.0, m[3]), chain3, chain4, chain4_list,
                                                b.sphere((0.0, 4.0, 1.0), 0.8, m[4])])

    # 3: lists nested in the world list with media inside them (dissolve in order: the medium draws at its place, hit.rs:969)
    b, m = base(3)
    fog_ball = b.sphere((1.2, 1.0, 0.0), 0.9, m[2])
    fog = b.constant_medium((0.3, 0.4, 0.9), 1.5, fog_ball)
    smoke = b.constant_medium((0.9, 0.9, 0.9), 0.8, b.translate((-1.8, 0.0, 0.0), b.rotate_y(30.0, b.rect_prism((0.0, 0.0, 0.0), (1.2, 1.4, 1.2), m[0]))))
    inner = b.hittable_list([fog_ball, fog, b.hittable_list([smoke, b.sphere((0.0, 0.5, 1.5), 0.5, m[1])])])
    yield "media_in_nested_lists", b, b.hittable_list([b.sphere((0.0, -500.0, 0.0), 500.0, m[3]), inner, b.sphere((0.0, 5.0, 0.0), 1.2, m[4])])


def _cam_cfg(rtsr, spp=6):
    cam = rtsr.Camera.new((1.0, 3.2, 8.5), (0.0, 0.9, 0.0), (0.0, 1.0, 0.0), 42.0, 1.5, 0.05, 8.0, 0.0, 1.0)
    cfg = rtsr.Config.new(1.5, 120, spp, 30, 4, seed=11, background=(0.35, 0.4, 0.55))
    return cam, cfg, rtsr.image_height(cfg)


def test_nested_containers_flatten_like_the_object_graph(rtsr, orc):
    cam, cfg, h = _cam_cfg(rtsr)
    for name, b, world in nested_worlds(rtsr):
        a1, r1 = orc.o1_render(b.graph_ptr(), world, cam, cfg, h, threads=8)
        for kw in ({}, {"max_leaf": 4}, {"reference_bvh": True, "bvh_seed": 3}):
            flat = b.flatten(world, **kw)
            a2, r2 = orc.o2_render(flat.arrays_ptr(), cam, cfg, h, threads=8)
            assert np.array_equal(a1, a2), "%s %r: %d pixels differ" % (name, kw, int((np.abs(a1 - a2).max(axis=2) > 0).sum()))
            assert np.array_equal(r1, r2), name
        assert a1.std() > 0.01, name  # the geometry is in view


def test_nested_bvh_becomes_one_tree(rtsr):
    for name, b, world in nested_worlds(rtsr):
        if name != "bvh_in_bvh":
            continue
        info = b.flatten(world).info()
        assert info["n_bvh"] == 1 and info["n_spheres"] == 1 + 1 + 7 + 12 + 9 and info["n_rects"] == 6
        assert info["n_refs"] == 1 + 7 + 12 + 9 + 6 and info["n_top_level"] == 2


@pytest.mark.gpu
def test_nested_containers_on_the_device(rtsr, orc):
    cam, cfg, h = _cam_cfg(rtsr, spp=16)
    for name, b, world in nested_worlds(rtsr):
        flat = b.flatten(world)
        a2, r2 = orc.o2_render(flat.arrays_ptr(), cam, cfg, h, threads=16)
        screen = flat.upload().render(cam, cfg)
        assert np.array_equal(screen.accum, a2), "%s: %d pixels differ" % (name, int((np.abs(screen.accum - a2).max(axis=2) > 0).sum()))
        assert np.array_equal(screen.rgb8, r2), name
