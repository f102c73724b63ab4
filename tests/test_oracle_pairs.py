"""O1 (literal object-graph restatement, reference BVH rule) vs O2 (product core over the product's
flattened arrays and SAH BVH), on the CPU.  Equality here means the flattener, the BVH builder and the
iterative traversal reproduce the reference's object-graph semantics bit for bit."""
import numpy as np
import pytest

# (name, scene id, width, image aspect, spp, options)
SCENES = [
    ("checkered", 0, 48, 16 / 9, 4, {}),
    ("two_perlin", 1, 48, 16 / 9, 4, {}),
    ("earth_image_texture", 2, 48, 16 / 9, 4, {}),
    ("simple_light", 3, 48, 16 / 9, 4, {}),
    ("cornell_box_xform", 4, 40, 1.0, 4, {}),
    ("cornell_smoke_media", 5, 40, 1.0, 4, {}),
    ("book2_final", 6, 48, 1.0, 4, {}),
    ("moving_test", 7, 48, 16 / 9, 4, {}),
    ("nested_lists", 9, 48, 16 / 9, 4, {}),
    ("triangle_test", 10, 48, 16 / 9, 4, {}),
    ("dragon_mesh_8k", 11, 64, 16 / 9, 2, {"mesh_triangles": 8000}),
    ("triangular_prism", 12, 40, 1.0, 4, {}),
    ("book1_head", 13, 64, 16 / 9, 4, {}),
    ("book1_canonical", 100, 64, 1.5, 4, {}),
    ("empty_world", 101, 16, 16 / 9, 2, {}),
]


def _scene(rtsr, sid, width, aspect, spp, opts, seed=5, **cfgkw):
    b = rtsr.Builder(1)
    world, cam, bg = b.get_world_cam(sid, **opts)
    cfg = rtsr.Config.new(aspect, width, spp, 50, 4, seed=seed, background=bg, **cfgkw)
    return b, world, cam, cfg, rtsr.image_height(cfg)


@pytest.mark.parametrize("name,sid,width,aspect,spp,opts", SCENES, ids=[s[0] for s in SCENES])
def test_flat_path_equals_literal_path(rtsr, orc, name, sid, width, aspect, spp, opts):
    b, world, cam, cfg, h = _scene(rtsr, sid, width, aspect, spp, opts)
    flat = b.flatten(world)
    a1, r1 = orc.o1_render(b.graph_ptr(), world, cam, cfg, h, threads=8)
    a2, r2 = orc.o2_render(flat.arrays_ptr(), cam, cfg, h, threads=8)
    assert np.isfinite(a1).all()
    assert np.array_equal(a1, a2), "%d pixels differ" % int((np.abs(a1 - a2).max(axis=2) > 0).sum())
    assert np.array_equal(r1, r2)


def tie_world(rtsr):
    """[rect A, BVH{rect B coplanar with A, sphere}, rect C, rect D coplanar with C]: exact ties between list
    entries of different kinds.  HittableList::hit (hit.rs:660-690) lets the later entry win: B over A, D over C."""
    b = rtsr.Builder(1)
    red, green = b.lambertian((0.8, 0.1, 0.1)), b.lambertian((0.1, 0.8, 0.1))
    blue, light = b.lambertian((0.1, 0.1, 0.8)), b.diffuse_light((3.0, 3.0, 3.0))
    inner = b.hittable_list([b.xz_rect(-2, 2, -2, 2, 0.0, green), b.sphere((0.0, 0.6, 0.0), 0.5, b.metal((0.8, 0.8, 0.8), 0.1))])
    world = b.hittable_list([b.xz_rect(-2, 2, -2, 2, 0.0, red), b.bvh_from_list(inner, 0.0, 1.0),
                             b.xz_rect(-3, 3, -3, 3, 4.0, blue), b.xz_rect(-3, 3, -3, -1.5, 4.0, light)])
    cam = rtsr.Camera.new((0.0, 2.0, 6.0), (0.0, 1.0, 0.0), (0.0, 1.0, 0.0), 50.0, 1.0, 0.0, 6.0, 0.0, 1.0)
    cfg = rtsr.Config.new(1.0, 96, 16, 20, 4, seed=5, background=(0.2, 0.2, 0.2))
    return b, world, cam, cfg, rtsr.image_height(cfg)


def test_list_ties_later_entry_wins(rtsr, orc):
    b, world, cam, cfg, h = tie_world(rtsr)
    flat = b.flatten(world)
    a1, r1 = orc.o1_render(b.graph_ptr(), world, cam, cfg, h, threads=8)
    a2, r2 = orc.o2_render(flat.arrays_ptr(), cam, cfg, h, threads=8)
    assert np.array_equal(a1, a2) and np.array_equal(r1, r2)
    floor_px = a1[24, 48]  # row 0 is the bottom row; (24, 48) looks at the floor in front of the sphere
    assert floor_px[1] > 4.0 * floor_px[0], "the BVH's green rectangle (later entry) must win the tie with the red one"
    ceiling_px = a1[92, 48] / 16.0  # the emitting patch (later) over the blue ceiling
    assert ceiling_px.min() > 1.0


def test_bvh_topology_does_not_matter(rtsr, orc):
    """Different reference-BVH axis streams (O1) and different SAH leaf sizes (O2): same image."""
    b, world, cam, cfg, h = _scene(rtsr, 100, 64, 1.5, 4, {})
    ref, _ = orc.o1_render(b.graph_ptr(), world, cam, cfg, h, threads=8, bvh_seed=1)
    other, _ = orc.o1_render(b.graph_ptr(), world, cam, cfg, h, threads=8, bvh_seed=99)
    assert np.array_equal(ref, other)
    for leaf in (1, 2, 4, 8):
        flat = b.flatten(world, max_leaf=leaf)
        a2, _ = orc.o2_render(flat.arrays_ptr(), cam, cfg, h, threads=8)
        assert np.array_equal(ref, a2), "max_leaf=%d" % leaf


def test_thread_count_and_seed(rtsr, orc):
    b, world, cam, cfg, h = _scene(rtsr, 13, 48, 16 / 9, 3, {})
    flat = b.flatten(world)
    one, _ = orc.o2_render(flat.arrays_ptr(), cam, cfg, h, threads=1)
    many, _ = orc.o2_render(flat.arrays_ptr(), cam, cfg, h, threads=7)
    assert np.array_equal(one, many)                       # scheduling cannot change a sample
    b2, world2, cam2, cfg2, _ = _scene(rtsr, 13, 48, 16 / 9, 3, {}, seed=6)
    flat2 = b2.flatten(world2)
    other, _ = orc.o2_render(flat2.arrays_ptr(), cam2, cfg2, h, threads=7)
    assert not np.array_equal(one, other)                  # the render seed does


def test_row_shards_tile_the_image(rtsr, orc):
    """Shards {j : (j / B) % N == r} rendered separately reassemble the unsharded image exactly."""
    b, world, cam, cfg, h = _scene(rtsr, 100, 40, 1.5, 2, {})
    flat = b.flatten(world)
    full, full8 = orc.o2_render(flat.arrays_ptr(), cam, cfg, h, threads=8)
    for n, block in [(2, 1), (3, 1), (8, 1), (2, 4), (5, 3)]:
        out = np.zeros_like(full)
        for r in range(n):
            part, _ = orc.o2_render(flat.arrays_ptr(), cam, cfg, h, shard=(r, n, block), threads=4)
            rows = [j for j in range(h) if (j // block) % n == r]
            assert part.shape[0] == len(rows) == rtsr.shard_rows(cfg, (r, n, block))
            out[rows] = part
        assert np.array_equal(out, full), (n, block)


def test_row_chunk_compat_quirk(rtsr, orc):
    """world.rs:1198-1202: with `threads` bands of floor(h/threads) rows the remainder rows stay black."""
    b, world, cam, cfg, h = _scene(rtsr, 100, 40, 1.5, 2, {}, row_chunk_compat=True)
    cfg.threads = 4  # h = 26 -> 4 bands of 6 rows, rows 24..25 never rendered
    flat = b.flatten(world)
    a1, r1 = orc.o1_render(b.graph_ptr(), world, cam, cfg, h)
    a2, r2 = orc.o2_render(flat.arrays_ptr(), cam, cfg, h)
    assert h == 26 and (a1[24:] == 0).all() and (r1[24:] == 0).all() and a1[:24].min() > 0
    assert np.array_equal(a1, a2) and np.array_equal(r1, r2)


def test_depth_exhaustion_and_background(rtsr, orc):
    """world.rs:64-67: depth is decremented first; an exhausted path contributes nothing (no background)."""
    b = rtsr.Builder(1)
    inside = b.sphere((0, 0, 0), 100.0, b.metal((1, 1, 1), 0.0))  # camera inside a mirror ball: never escapes
    cam = rtsr.Camera.new((0, 0, 0), (0, 0, -1), (0, 1, 0), 40.0, 1.0, 0.0, 1.0, 0.0, 1.0)
    cfg = rtsr.Config.new(1.0, 8, 2, 7, 1, background=(0.7, 0.8, 1.0))
    a1, _ = orc.o1_render(b.graph_ptr(), inside, cam, cfg, 8)
    flat_inside = b.flatten(inside)
    a2, _ = orc.o2_render(flat_inside.arrays_ptr(), cam, cfg, 8)
    assert (a1 == 0).all() and (a2 == 0).all()
    empty = b.hittable_list()
    flat_empty = b.flatten(empty)
    a1, _ = orc.o1_render(b.graph_ptr(), empty, cam, cfg, 8)
    a2, _ = orc.o2_render(flat_empty.arrays_ptr(), cam, cfg, 8)
    assert np.array_equal(a1, a2) and np.allclose(a1, np.array([0.7, 0.8, 1.0]) * 2)


def test_statistical_agreement_between_scene_seeds(rtsr, orc):
    """Different scene seeds give different sphere layouts but the same kind of image (sanity of the generator)."""
    means = []
    for seed in (1, 2):
        b = rtsr.Builder(seed)
        world, cam, bg = b.get_world_cam(100)
        cfg = rtsr.Config.new(1.5, 48, 4, 50, 4, background=bg)
        flat = b.flatten(world)
        a, _ = orc.o2_render(flat.arrays_ptr(), cam, cfg, 32, threads=8)
        means.append(a.mean() / 4)
    assert means[0] != means[1] and abs(means[0] - means[1]) < 0.1


# ---- objects shared between containers (the reference holds every object as an Arc: one Triangle, or one BvhNode,
# may sit in several lists / under several Translates) and degenerate boxes ---------------------------------------
def _tri_fan(b, mat, n, z0=0.0, dx=0.0):
    """n small non-coplanar triangles on a grid facing +z (slightly tilted so that no bounding box is flat)."""
    tris = []
    for k in range(n):
        x, y = (k % 8) * 1.0 - 4.0 + dx, (k // 8) * 1.0 - 2.0
        tris.append(b.triangle((x, y, z0 + 0.05 * (k % 3)), (x + 0.9, y, z0), (x, y + 0.9, z0 + 0.1), mat))
    return tris


def _shared_cam_cfg(rtsr):
    cam = rtsr.Camera.new((0.0, 0.5, 14.0), (0.0, 0.5, 0.0), (0.0, 1.0, 0.0), 50.0, 1.5, 0.0, 14.0, 0.0, 1.0)
    cfg = rtsr.Config.new(1.5, 96, 4, 8, 4, seed=3, background=(0.6, 0.7, 0.9))
    return cam, cfg, rtsr.image_height(cfg)


def shared_worlds(rtsr):
    """(name, builder, world) for every sharing shape the flattener must keep apart."""
    out = []
    # (a) a triangle of a BVH is ALSO a plain entry under a Translate, listed after the BVH
    b = rtsr.Builder(1)
    m1, m2 = b.lambertian((0.8, 0.2, 0.2)), b.metal((0.7, 0.7, 0.7), 0.1)
    tris = _tri_fan(b, m1, 24)
    world = b.hittable_list([b.bvh_from_list(b.hittable_list(tris), 0.0, 1.0), b.translate((0.0, 5.0, 0.0), tris[6]),
                             b.sphere((0.0, -1001.0, 0.0), 998.0, m2)])
    out.append(("prim_shared_after_bvh", b, world))
    # (a') the shared triangle is emitted BEFORE the BVH that also holds it
    b = rtsr.Builder(1)
    m1 = b.lambertian((0.2, 0.8, 0.2))
    tris = _tri_fan(b, m1, 24)
    world = b.hittable_list([b.translate((0.0, 5.0, 0.0), tris[3]), tris[17], b.bvh_from_list(b.hittable_list(tris), 0.0, 1.0)])
    out.append(("prim_shared_before_bvh", b, world))
    # (b) two BVHs over overlapping triangle sets
    b = rtsr.Builder(1)
    m1 = b.lambertian((0.2, 0.2, 0.8))
    tris = _tri_fan(b, m1, 40)
    world = b.hittable_list([b.bvh_from_list(b.hittable_list(tris[:30]), 0.0, 1.0),
                             b.translate((0.3, 0.2, 1.0), b.bvh_from_list(b.hittable_list(tris[10:]), 0.0, 1.0))])
    out.append(("overlapping_bvhs", b, world))
    # (c) ONE BVH handle instanced under two transforms (and once plainly)
    b = rtsr.Builder(1)
    m1 = b.lambertian((0.8, 0.8, 0.2))
    bvh = b.bvh_from_list(b.hittable_list(_tri_fan(b, m1, 24)), 0.0, 1.0)
    world = b.hittable_list([bvh, b.translate((0.0, 3.2, 0.0), bvh), b.translate((0.0, -3.2, 0.0), b.rotate_y(20.0, bvh))])
    out.append(("instanced_bvh", b, world))
    # (d) the same triangle listed twice inside one BVH
    b = rtsr.Builder(1)
    m1 = b.lambertian((0.8, 0.2, 0.8))
    tris = _tri_fan(b, m1, 16)
    world = b.hittable_list([b.bvh_from_list(b.hittable_list(tris + [tris[5], tris[5]]), 0.0, 1.0)])
    out.append(("duplicate_in_bvh", b, world))
    return out


@pytest.mark.parametrize("reference_bvh", [False, True], ids=["sah", "reference_rule"])
def test_shared_objects_flatten_like_the_object_graph(rtsr, orc, reference_bvh):
    cam, cfg, h = _shared_cam_cfg(rtsr)
    for name, b, world in shared_worlds(rtsr):
        a1, r1 = orc.o1_render(b.graph_ptr(), world, cam, cfg, h, threads=8)
        flat = b.flatten(world, reference_bvh=reference_bvh, bvh_seed=7)
        a2, r2 = orc.o2_render(flat.arrays_ptr(), cam, cfg, h, threads=8)
        assert np.array_equal(a1, a2), "%s: %d pixels differ" % (name, int((np.abs(a1 - a2).max(axis=2) > 0).sum()))
        assert np.array_equal(r1, r2), name
        assert a1.std() > 0.01, name  # the geometry is in view


def axis_aligned_world(rtsr, as_bvh):
    """Triangles lying exactly in coordinate planes (zero-thickness boxes, hit.rs:164-177 pads nothing) among tilted
    ones.  As a plain HittableList the reference hits every one of them; inside a BvhNode whether it does depends on
    its random split axes (a flat union box is never entered, aabb.rs:58) -- this build always hits them."""
    b = rtsr.Builder(1)
    m = [b.lambertian(c) for c in ((0.8, 0.2, 0.2), (0.2, 0.8, 0.2), (0.2, 0.2, 0.8))]
    tris = []
    for k in range(12):
        x = k * 0.8 - 4.8
        tris.append(b.triangle((x, -1.0, 0.0), (x + 0.7, -1.0, 0.0), (x, 0.2, 0.0), m[k % 3]))          # in the plane z = 0
        tris.append(b.triangle((x, 0.5, -1.0), (x + 0.7, 0.5, 0.3), (x, 0.5, 0.3), m[(k + 1) % 3]))      # in the plane y = 0.5
        tris.append(b.triangle((x, 1.0, 0.0), (x + 0.7, 1.2, 0.4), (x + 0.1, 2.0, -0.3), m[(k + 2) % 3]))  # tilted
    lst = b.hittable_list(tris)
    world = b.hittable_list([b.bvh_from_list(lst, 0.0, 1.0)]) if as_bvh else lst
    return b, world


def test_axis_aligned_triangles(rtsr, orc):
    cam = rtsr.Camera.new((0.5, 3.0, 9.0), (0.0, 0.3, 0.0), (0.0, 1.0, 0.0), 55.0, 1.5, 0.0, 9.0, 0.0, 1.0)
    cfg = rtsr.Config.new(1.5, 96, 4, 8, 4, seed=3, background=(0.6, 0.7, 0.9))
    h = rtsr.image_height(cfg)
    b0, list_world = axis_aligned_world(rtsr, as_bvh=False)
    expect, expect8 = orc.o1_render(b0.graph_ptr(), list_world, cam, cfg, h, threads=8)  # literal list scan: no boxes involved
    b1, bvh_world = axis_aligned_world(rtsr, as_bvh=True)
    for kw in ({}, {"max_leaf": 1}, {"max_leaf": 4}, {"reference_bvh": True, "bvh_seed": 1}, {"reference_bvh": True, "bvh_seed": 2}):
        flat = b1.flatten(bvh_world, **kw)
        a2, r2 = orc.o2_render(flat.arrays_ptr(), cam, cfg, h, threads=8)
        assert np.array_equal(expect, a2), "%r: %d pixels differ" % (kw, int((np.abs(expect - a2).max(axis=2) > 0).sum()))
        assert np.array_equal(expect8, r2)
