"""The reference's own unit tests (src/vec3.rs:343-428, all nine) and the hand-derived known-answer
tests of SURVEY.md section 4, run against BOTH CPU restatements:
  "o1"   oracle/o1_literal.cpp  (literal object-graph restatement, its own Vec3)
  "core" the product's shared host/device core compiled for the host (what the kernels inline)
These are the only pins the reference's repository provides for this path.
"""
import math

import pytest

SIDES = ["o1", "core"]


@pytest.mark.parametrize("side", SIDES)
def test_general_vec3_stuff(orc, side):  # vec3.rs:348-363
    v1, v2 = (2, 2, 1), (5, 7, 4.1)
    r = orc.vec3_ops(side, v1, v2, 1.0)
    assert r["length_squared"] == 9.0
    assert r["length"] == 3.0
    assert r["add"] == (7.0, 9.0, 5.1)
    assert orc.vec3_ops(side, v2, v1, 1.0)["neg"] == (-5.0, -7.0, -4.1)


@pytest.mark.parametrize("side", SIDES)
def test_add_assign(orc, side):  # vec3.rs:366-370
    assert orc.vec3_ops(side, (3, 2, 1), (3, 2, 1), 1.0)["add"] == (6.0, 4.0, 2.0)


@pytest.mark.parametrize("side", SIDES)
def test_mul_assign(orc, side):  # vec3.rs:373-379
    assert orc.vec3_ops(side, (3, 2, 1), (0, 0, 0), 5)["mul_t"] == (15.0, 10.0, 5.0)
    assert orc.vec3_ops(side, (15, 10, 5), (0, 0, 0), 2.5)["mul_t"] == (37.5, 25.0, 12.5)


@pytest.mark.parametrize("side", SIDES)
def test_div_assign(orc, side):  # vec3.rs:382-386  (v /= 3  ==  v *= 1/3)
    r = orc.vec3_ops(side, (27, 9, 3), (0, 0, 0), 1.0 / 3.0)["mul_t"]
    assert r == (27 * (1.0 / 3.0), 9 * (1.0 / 3.0), 3 * (1.0 / 3.0)) == (9.0, 3.0, 1.0)


@pytest.mark.parametrize("side", SIDES)
def test_f64_mul(orc, side):  # vec3.rs:389-392
    assert orc.vec3_ops(side, (5, 10, 15), (0, 0, 0), 0.5)["mul_t"] == (2.5, 5.0, 7.5)


@pytest.mark.parametrize("side", SIDES)
def test_div_f64(orc, side):  # vec3.rs:395-398
    assert orc.vec3_ops(side, (5, 10, 15), (0, 0, 0), 5)["div_t"] == (1.0, 2.0, 3.0)


@pytest.mark.parametrize("side", SIDES)
def test_cross(orc, side):  # vec3.rs:401-406
    assert orc.vec3_ops(side, (2, 3, 4), (5, 6, 7), 1.0)["cross"] == (-3.0, 6.0, -3.0)


@pytest.mark.parametrize("side", SIDES)
def test_vec3_iter_and_zip(orc, side):  # vec3.rs:409-427: iteration order is x, y, z
    r = orc.vec3_ops(side, (5, 6, 7), (7, 8, 9), 1.0)
    assert r["mul"] == (35.0, 48.0, 63.0)  # component pairing (5,7), (6,8), (7,9)
    assert r["dot"] == 5 * 7 + 6 * 8 + 7 * 9


# ------------------------------------------------------------------ SURVEY.md section 4 KATs
@pytest.mark.parametrize("side", SIDES)
def test_tone_map(orc, side):  # vec3.rs:10,89-107, mutil.rs:1-9
    spp = 16
    assert orc.tone_map(side, (spp, spp, spp), spp) == (255, 255, 255)
    assert orc.tone_map(side, (0.25 * spp,) * 3, spp) == (127, 127, 127)  # (255.9 * 0.5) as i32
    assert orc.tone_map(side, (0, 0, 0), spp) == (0, 0, 0)
    assert orc.tone_map(side, (10 * spp, 2 * spp, 1.0001 * spp), spp) == (255, 255, 255)
    assert orc.tone_map(side, (float("nan"), -1.0, float("inf")), spp) == (0, 0, 255)  # NaN -> 0 (saturating cast)


@pytest.mark.parametrize("side", SIDES)
def test_sphere_uv(orc, side):  # hit.rs:195-200
    assert orc.sphere_uv(side, (1, 0, 0)) == pytest.approx((0.5, 0.5), abs=1e-15)
    assert orc.sphere_uv(side, (0, 1, 0)) == pytest.approx((0.5, 1.0), abs=1e-15)
    assert orc.sphere_uv(side, (0, 0, 1)) == pytest.approx((0.25, 0.5), abs=1e-15)
    u, v = orc.sphere_uv(side, (-1, 0, 0))
    assert v == pytest.approx(0.5, abs=1e-15) and (u == pytest.approx(0.0, abs=1e-15) or u == pytest.approx(1.0, abs=1e-15))


@pytest.mark.parametrize("side", SIDES)
def test_dielectric_reflectance(orc, side):  # hit.rs:1095-1099
    assert orc.reflectance(side, 1.0, 1.5) == pytest.approx(0.04, abs=1e-16)
    assert orc.reflectance(side, 0.0, 1.5) == 1.0
    x = 0.3
    r0 = ((1.0 - 1.5) / (1.0 + 1.5)) ** 2
    assert orc.reflectance(side, 1.0 - x, 1.5) == r0 + (1.0 - r0) * (x * ((x * x) * (x * x)))  # powi(x,5)


@pytest.mark.parametrize("side", SIDES)
def test_refract_and_reflect(orc, side):  # vec3.rs:64-66, 116-121
    assert orc.refract(side, (0, -1, 0), (0, 1, 0), 1 / 1.5) == (0.0, -1.0, 0.0)
    assert orc.reflect(side, (1, -1, 0), (0, 1, 0)) == (1.0, 1.0, 0.0)


@pytest.mark.parametrize("side", SIDES)
def test_aabb_hit(orc, side):  # aabb.rs:46-60
    mn, mx, o, d = (1, 1, 1), (2, 2, 2), (0, 0, 0), (1, 1, 1)
    assert orc.aabb_hit(side, mn, mx, o, d, 0.001, float("inf")) is True
    assert orc.aabb_hit(side, mn, mx, o, d, 0.001, 1.0) is False  # t_max <= t_min at exactly 1.0
    assert orc.aabb_hit(side, mn, mx, o, (-1, -1, -1), 0.001, float("inf")) is False
    # a zero direction component: inv_d = inf, NaN comparisons keep the old bounds (aabb.rs:48-58)
    assert orc.aabb_hit(side, (-1, -1, -1), (1, 1, 1), (0, 0, -5), (0, 0, 1), 0.001, float("inf")) is True
    assert orc.aabb_hit(side, (-1, -1, -1), (1, 1, 1), (2, 0, -5), (0, 0, 1), 0.001, float("inf")) is False


def _both_hits(rtsr, orc, build, o, d, **kw):
    b = rtsr.Builder(1)
    h = build(b)
    flat = b.flatten(h)
    return orc.o1_hit(b.graph_ptr(), h, o, d, **kw), orc.core_world_hit(flat.arrays_ptr(), o, d, **kw)


def test_sphere_hit_kat(rtsr, orc):  # hit.rs:204-236
    for rec in _both_hits(rtsr, orc, lambda b: b.sphere((0, 0, -1), 0.5, b.lambertian((0.5, 0.5, 0.5))), (0, 0, 0), (0, 0, -1)):
        assert rec["t"] == 0.5 and rec["p"] == (0.0, 0.0, -0.5) and rec["normal"] == (0.0, 0.0, 1.0) and rec["front_face"]
    # from inside: far root, normal flipped against the ray
    for rec in _both_hits(rtsr, orc, lambda b: b.sphere((0, 0, 0), 2.0, b.dielectric(1.5)), (0, 0, 0), (0, 0, -1)):
        assert rec["t"] == 2.0 and rec["normal"] == (0.0, 0.0, 1.0) and not rec["front_face"]


def test_triangle_hit_kat(rtsr, orc):  # hit.rs:96-107,111-162, world.rs:667-672 (scene 10's triangle)
    build = lambda b: b.triangle((0, 5, 0), (5, 0, 0), (0, 0, 0), b.lambertian((1, 0, 0)))
    for rec in _both_hits(rtsr, orc, build, (1, 1, 20), (0, 0, -1)):
        assert rec["t"] == 20.0 and rec["normal"] == (0.0, 0.0, 1.0) and not rec["front_face"]
        assert rec["u"] == 1.0 and rec["v"] == 1.0
    for rec in _both_hits(rtsr, orc, build, (4, 4, 20), (0, 0, -1)):  # outside the hypotenuse
        assert rec is None
    for rec in _both_hits(rtsr, orc, build, (1, 1, 20), (1, 0, -0.00001)):  # |n.d| < 1e-4: treated as parallel
        assert rec is None


def test_rect_and_moving_sphere_kat(rtsr, orc):
    mat = lambda b: b.lambertian((0.5, 0.5, 0.5))
    for rec in _both_hits(rtsr, orc, lambda b: b.xz_rect(-1, 1, -2, 2, 3.0, mat(b)), (0.5, 0, 1), (0, 1, 0)):  # hit.rs:541-566
        assert rec["t"] == 3.0 and rec["normal"] == (0.0, -1.0, 0.0) and not rec["front_face"]
    for rec in _both_hits(rtsr, orc, lambda b: b.yz_rect(-1, 1, -2, 2, 3.0, mat(b)), (0, 0.5, 1), (1, 0, 0)):  # hit.rs:606-631
        assert rec["t"] == 3.0 and rec["normal"] == (-1.0, 0.0, 0.0)
    # MovingSphere centre at time t (hit.rs:275-278): c0 + ((t - t0)/(t1 - t0)) * (c1 - c0); u = v = 0
    build = lambda b: b.moving_sphere((0, 0, -5), (0, 4, -5), 0.0, 2.0, 1.0, mat(b))
    for rec in _both_hits(rtsr, orc, build, (0, 2, 0), (0, 0, -1), time=1.0):
        assert rec["t"] == 4.0 and rec["u"] == 0.0 and rec["v"] == 0.0
    for rec in _both_hits(rtsr, orc, build, (0, 2, 0), (0, 0, -1), time=0.0):
        assert rec is None


def test_list_tie_later_object_wins(rtsr, orc):  # hit.rs:676-680: strict rejection -> equal t replaces
    def build(b):
        lst = b.hittable_list()
        b.list_add(lst, b.xz_rect(-1, 1, -1, 1, 5.0, b.metal((1, 1, 1), 0.0)))
        b.list_add(lst, b.xz_rect(-2, 2, -2, 2, 5.0, b.diffuse_light((4, 4, 4))))
        return lst
    b = rtsr.Builder(1)
    world = build(b)
    flat = b.flatten(world)
    cam = rtsr.Camera.new((0, 0, 0), (0, 1, 0), (0, 0, 1), 20.0, 1.0, 0.0, 5.0, 0.0, 1.0)
    cfg = rtsr.Config.new(1.0, 8, 2, 5, 1, background=(0, 0, 0))
    a1, _ = orc.o1_render(b.graph_ptr(), world, cam, cfg, 8)
    a2, _ = orc.o2_render(flat.arrays_ptr(), cam, cfg, 8)
    assert (a1 == a2).all()
    assert a1[4, 4, 0] == 8.0  # the light (added last) wins the coplanar tie: 2 spp x emitted 4.0


def test_image_height_formula(rtsr):  # world.rs:1192
    for width, aspect, want in [(800, 1.5, 533), (200, 1.5, 133), (1920, 16 / 9, 1080), (3840, 16 / 9, 2160), (600, 1.6, 375)]:
        assert rtsr.image_height(rtsr.Config.new(aspect, width, 1, 1, 1)) == want
