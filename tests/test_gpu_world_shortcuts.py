"""GPU parity for the world-list shortcuts of k_trace_world (trace_world.inc): one quadratic per sphere that is asked more
than once (a medium's boundary; a glass ball followed by the medium inside the very same sphere), media and moving spheres
skipped by a conservative box.  Each world below turns exactly one of them on or OFF by a detail of the scene; all must
equal the literal oracle O1 bit for bit (hit.rs:204-222 Sphere::hit, 282-300 MovingSphere::hit, 948-990 ConstantMedium::hit).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _room(b):
    """A lit box of rectangles around the action, so that paths bounce and media are entered from every side."""
    white = b.lambertian((0.73, 0.73, 0.73))
    return [b.xz_rect(-40, 40, -40, 40, 30.0, b.diffuse_light((4, 4, 4))), b.xz_rect(-60, 60, -60, 60, -20.0, white),
            b.xy_rect(-60, 60, -20, 30, 60.0, white), b.yz_rect(-20, 30, -60, 60, -60.0, b.lambertian((0.65, 0.05, 0.05))),
            b.yz_rect(-20, 30, -60, 60, 60.0, b.lambertian((0.12, 0.45, 0.15)))]


def _cluster(b):
    """A BVH entry, so that the world takes k_trace_world and not the one-BVH kernels."""
    rng = np.random.default_rng(11)
    m = b.lambertian((0.5, 0.5, 0.7))
    return b.bvh_from_list(b.hittable_list([b.sphere(tuple(rng.uniform(-30, 30, 3) * (1, 0.3, 1) + (0, -12, 0)), 2.0, m) for _ in range(40)]), 0.0, 1.0)


def _worlds(b):
    glass = b.dielectric(1.5)
    out = {}
    # the glass ball and the medium inside the very same sphere, adjacent in the list (Book-2's pattern): WD_PAIR
    out["pair"] = [b.sphere((0, 0, 0), 12.0, glass), b.constant_medium((0.2, 0.4, 0.9), 0.2, b.sphere((0, 0, 0), 12.0, glass))]
    # the same two objects with another object between them: no pair, the medium alone solves once
    out["not_adjacent"] = [b.sphere((0, 0, 0), 12.0, glass), b.sphere((25, 0, 0), 5.0, b.metal((0.8, 0.8, 0.9), 0.3)),
                           b.constant_medium((0.2, 0.4, 0.9), 0.2, b.sphere((0, 0, 0), 12.0, glass))]
    # a boundary that differs from the glass ball in the last bit of the radius: must not pair
    out["nearly_same"] = [b.sphere((0, 0, 0), 12.0, glass), b.constant_medium((0.9, 0.4, 0.2), 0.1, b.sphere((0, 0, 0), float(np.nextafter(12.0, 13.0)), glass))]
    # medium first, ball second: the pair rule is about order
    out["medium_first"] = [b.constant_medium((1, 1, 1), 0.05, b.sphere((0, 0, 0), 12.0, glass)), b.sphere((0, 0, 0), 12.0, glass)]
    # the camera inside a huge medium (Book-2's fog): every ray has rec1.t < 0
    out["fog_around_everything"] = [b.sphere((0, 0, 0), 500.0, glass), b.constant_medium((1, 1, 1), 0.003, b.sphere((0, 0, 0), 500.0, glass))]
    # a medium under a translation keeps the general path
    out["moved_medium"] = [b.constant_medium((0.1, 0.8, 0.3), 0.15, b.translate((10, 0, 5), b.sphere((0, 0, 0), 9.0, glass)))]
    # moving spheres: one whose own (time0, time1) covers the shutter (boxed), one whose interval does not (never boxed)
    out["moving_inside_shutter"] = [b.moving_sphere((-10, 0, 0), (10, 6, 0), 0.0, 1.0, 6.0, b.lambertian((0.7, 0.3, 0.1)))]
    out["moving_outside_shutter"] = [b.moving_sphere((-10, 0, 0), (10, 6, 0), 0.25, 0.5, 6.0, b.lambertian((0.7, 0.3, 0.1)))]
    return out


@pytest.mark.parametrize("name", ["pair", "not_adjacent", "nearly_same", "medium_first", "fog_around_everything", "moved_medium",
                                  "moving_inside_shutter", "moving_outside_shutter"])
def test_world_list_shortcuts_equal_the_literal_oracle(rtsr, orc, name):
    b = rtsr.Builder(3)
    world = b.hittable_list([_cluster(b)] + _worlds(b)[name] + _room(b))
    cam = rtsr.Camera.new((0, 5, -55), (0, 0, 0), (0, 1, 0), 50.0, 1.25, 0.0, 10.0, 0.0, 1.0)
    cfg = rtsr.Config.new(1.25, 60, 6, 50, 4, seed=9, background=(0.05, 0.05, 0.08))
    flat = b.flatten(world)
    scene = flat.upload()
    st = scene.render_device(cam, cfg, want_stats=True)
    assert rtsr.trace_kernel_name(st.trace_kernel) == "k_trace_world"
    screen = scene.render(cam, cfg)
    h = rtsr.image_height(cfg)
    a1, r1 = orc.o1_render(b.graph_ptr(), world, cam, cfg, h)
    assert np.array_equal(screen.accum, a1) and np.array_equal(screen.rgb8, r1)
    a2, r2 = orc.o2_render(flat.arrays_ptr(), cam, cfg, h)
    assert np.array_equal(screen.accum, a2)
