"""Host-side logic of the product that runs without a GPU: flattener, SAH BVH builder, scene catalogue,
PPM writer/reader (screen.rs:40-95), PLY loader (model.rs:13-62), shard arithmetic."""
import numpy as np
import pytest


def test_flat_info_book1(rtsr):
    b = rtsr.Builder(1)
    world, cam, bg = b.get_world_cam(rtsr.SCENE_BOOK1_CANONICAL)
    info = b.flatten(world).info()
    n = info["n_spheres"]
    assert 470 <= n <= 488                      # 1 ground + <= 484 small + 3 big (world.rs:107-160)
    assert info["n_refs"] == n and info["n_bvh"] == 1 and info["n_top_level"] == 1
    assert info["n_nodes"] >= n // 2 - 1 and 6 <= info["max_stack"] <= 40
    b2 = rtsr.Builder(1)
    world2, _, _ = b2.get_world_cam(rtsr.SCENE_BOOK1_HEAD)
    info2 = b2.flatten(world2).info()
    assert info2["n_moving_spheres"] > 300 and info2["n_spheres"] > 50  # choose_mat < 0.8 -> MovingSphere (world.rs:128)
    assert info2["n_textures"] >= 3             # checker + its two solid colours, plus one per diffuse sphere


def test_flat_info_book2_and_dragon(rtsr):
    b = rtsr.Builder(1)
    world, cam, bg = b.get_world_cam(rtsr.SCENE_BOOK2_FINAL)
    info = b.flatten(world).info()
    assert info["n_rects"] == 400 * 6 + 1       # 400 RectPrisms x 6 sides + the light (world.rs:500-523)
    assert info["n_spheres"] == 1000 + 6 + 2    # instanced cluster + 6 plain + 2 medium boundaries
    assert info["n_moving_spheres"] == 1 and info["n_top_level"] == 12 and info["n_bvh"] == 2
    assert info["n_perlins"] == 1 and info["n_images"] == 1 and info["n_texels"] == 1024 * 512
    b = rtsr.Builder(1)
    world, cam, bg = b.get_world_cam(rtsr.SCENE_STANFORD_DRAGON, mesh_triangles=20000)
    info = b.flatten(world).info()
    assert 19000 <= info["n_triangles"] <= 20000 and info["n_rects"] == 7 and info["n_top_level"] == 8


@pytest.mark.parametrize("sid,opts", [(100, {}), (13, {}), (6, {}), (11, {"mesh_triangles": 30000})])
@pytest.mark.parametrize("leaf", [1, 2, 4, 8])
def test_bvh_structure(rtsr, orc, sid, opts, leaf):
    """Every BVH primitive sits in exactly one leaf, child boxes nest, and max_stack covers the depth."""
    b = rtsr.Builder(3)
    world, _, _ = b.get_world_cam(sid, **opts)
    flat = b.flatten(world, max_leaf=leaf)
    rc, depth = orc.audit_flat(flat.arrays_ptr())
    assert rc == 0, "audit code %d" % rc
    assert depth <= flat.info()["max_stack"] + 1


def test_bvh_degenerate_inputs(rtsr, orc):
    """Coincident and single-object inputs (span 1 / span 2 cases of bvh.rs:53-63)."""
    b = rtsr.Builder(1)
    m = b.lambertian((0.5, 0.5, 0.5))
    same = b.bvh_from_list(b.hittable_list([b.sphere((0, 0, -5), 1.0, m) for _ in range(9)]), 0, 1)
    one = b.bvh_from_list(b.hittable_list([b.sphere((0, 0, -5), 1.0, m)]), 0, 1)
    two = b.bvh_from_list(b.hittable_list([b.sphere((0, 0, -5), 1.0, m), b.sphere((3, 0, -5), 1.0, m)]), 0, 1)
    cam = rtsr.Camera.new((0, 0, 0), (0, 0, -1), (0, 1, 0), 60.0, 1.0, 0.0, 1.0, 0.0, 1.0)
    cfg = rtsr.Config.new(1.0, 12, 2, 10, 1)
    for world in (same, one, two):
        flat = b.flatten(world)
        assert orc.audit_flat(flat.arrays_ptr())[0] == 0
        a1, _ = orc.o1_render(b.graph_ptr(), world, cam, cfg, 12)
        a2, _ = orc.o2_render(flat.arrays_ptr(), cam, cfg, 12)
        assert np.array_equal(a1, a2)


def test_ppm_writer_format(rtsr, tmp_path):  # screen.rs:40-48, vec3.rs:109-114
    w, h = 3, 2
    rgb = np.arange(w * h * 3, dtype=np.uint8).reshape(h, w, 3) * 7
    path = tmp_path / "out.ppm"
    rtsr.Screen(w, h, rgb).write_to_ppm_file(str(path))
    lines = path.read_text().split("\n")
    assert lines[:3] == ["P3", "3 2", "255"] and lines[-1] == ""
    assert len(lines) == 3 + w * h + 1
    assert lines[3] == "%d %d %d" % tuple(rgb[h - 1, 0])      # first pixel line = top row (j = h-1), column 0
    assert lines[3 + w] == "%d %d %d" % tuple(rgb[0, 0])      # bottom row comes last
    assert all("." not in ln for ln in lines[3:-1])            # integer-valued, no decimal point


def test_ppm_reader_feeds_image_texture(rtsr, orc, tmp_path):  # screen.rs:61-95, texture.rs:102-121
    w, h = 4, 2
    rgb = (np.arange(w * h * 3).reshape(h, w, 3) * 9 % 256).astype(np.uint8)
    path = tmp_path / "tex.ppm"
    path.write_text("P3\n%d %d\n255\n" % (w, h) + "".join("%d %d %d\n" % tuple(p) for row in rgb for p in row))
    b = rtsr.Builder(1)
    world = b.sphere((0, 0, 0), 2.0, b.lambertian(b.image_from_ppm(str(path))))
    b2 = rtsr.Builder(1)
    world2 = b2.sphere((0, 0, 0), 2.0, b2.lambertian(b2.image_from_texels(rgb.astype(np.float64))))
    cam = rtsr.Camera.new((0, 0, 8), (0, 0, 0), (0, 1, 0), 40.0, 1.0, 0.0, 8.0, 0.0, 1.0)
    cfg = rtsr.Config.new(1.0, 16, 2, 4, 1)
    f1, f2 = b.flatten(world), b2.flatten(world2)
    a, _ = orc.o2_render(f1.arrays_ptr(), cam, cfg, 16)
    a2, _ = orc.o2_render(f2.arrays_ptr(), cam, cfg, 16)
    o1, _ = orc.o1_render(b.graph_ptr(), world, cam, cfg, 16)
    assert np.array_equal(a, a2) and np.array_equal(a, o1) and f1.info()["n_texels"] == w * h


def test_ply_loader(rtsr, orc, tmp_path):  # model.rs:13-62
    ply = tmp_path / "tetra.ply"
    ply.write_text("ply\nformat ascii 1.0\ncomment made by hand\nelement vertex 4\nproperty float x\nproperty float y\n"
                   "property float z\nelement face 4\nproperty list uchar int vertex_indices\nend_header\n"
                   "0 0 0\n1 0 0\n0 1 0\n0 0 1\n3 0 1 2\n3 0 1 3\n3 0 2 3\n3 1 2 3\n")
    b = rtsr.Builder(1)
    tris = b.triangle_model(str(ply), 100.0)           # scale multiplies every coordinate (model.rs:44-46)
    world = b.bvh_from_list(tris, 0.0, 1.0)
    flat = b.flatten(world)
    info = flat.info()
    assert info["n_triangles"] == 4 and info["n_materials"] == 1
    b2 = rtsr.Builder(1)
    verts = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], dtype=np.float64) * 100.0
    faces = np.array([[0, 1, 2], [0, 1, 3], [0, 2, 3], [1, 2, 3]])
    world2 = b2.bvh_from_list(b2.triangle_mesh(verts, faces, b2.lambertian((0.2, 0.2, 0.2))), 0.0, 1.0)
    cam = rtsr.Camera.new((300, 200, 400), (20, 20, 20), (0, 1, 0), 30.0, 1.0, 0.0, 10.0, 0.0, 1.0)
    cfg = rtsr.Config.new(1.0, 24, 2, 8, 1)
    flat2 = b2.flatten(world2)
    a, _ = orc.o2_render(flat.arrays_ptr(), cam, cfg, 24)
    a2, _ = orc.o2_render(flat2.arrays_ptr(), cam, cfg, 24)
    o1, _ = orc.o1_render(b.graph_ptr(), world, cam, cfg, 24)
    assert np.array_equal(a, a2) and np.array_equal(a, o1)
    assert len(np.unique(a.reshape(-1, 3), axis=0)) >= 3  # the tetrahedron is in view (grey 0.2 albedo vs sky)


def test_shard_rows(rtsr):
    cfg = rtsr.Config.new(1.5, 800, 1, 1, 1)
    assert rtsr.image_height(cfg) == 533
    for n in (1, 2, 4, 8):
        rows = [rtsr.shard_rows(cfg, (r, n, 1)) for r in range(n)]
        assert sum(rows) == 533 and max(rows) - min(rows) <= 1
    assert [rtsr.shard_rows(cfg, (r, 8, 8)) for r in range(8)] == [72, 72, 69, 64, 64, 64, 64, 64]


def test_scene_catalogue_is_deterministic(rtsr):
    infos = []
    for _ in range(2):
        b = rtsr.Builder(42)
        world, cam, bg = b.get_world_cam(rtsr.SCENE_BOOK1_HEAD)
        info = b.flatten(world).info()
        info.pop("bvh_build_ms")  # a timer, not a property of the scene
        infos.append((info, bytes(cam), bg))
    assert infos[0] == infos[1]
    b = rtsr.Builder(43)
    world, _, _ = b.get_world_cam(rtsr.SCENE_BOOK1_HEAD)
    assert b.flatten(world).info() != infos[0][0]


def test_reference_bvh_rule_same_image_more_box_tests(rtsr, orc):
    """building the BVH with the reference's rule (bvh.rs:14-83) instead of SAH is an A/B switch:
    identical image, very different traversal statistics."""
    b = rtsr.Builder(1)
    world, cam, bg = b.get_world_cam(rtsr.SCENE_BOOK1_CANONICAL)
    cfg = rtsr.Config.new(1.5, 48, 3, 50, 4, seed=3, background=bg)
    f_sah = b.flatten(world)
    f_ref = b.flatten(world, reference_bvh=True, bvh_seed=5)
    f_ref2 = b.flatten(world, reference_bvh=True, bvh_seed=6)
    a, _, c_sah = orc.o2_render(f_sah.arrays_ptr(), cam, cfg, 32, threads=8, counters=True)
    a_ref, _, c_ref = orc.o2_render(f_ref.arrays_ptr(), cam, cfg, 32, threads=8, counters=True)
    a_ref2, _ = orc.o2_render(f_ref2.arrays_ptr(), cam, cfg, 32, threads=8)
    assert np.array_equal(a, a_ref) and np.array_equal(a, a_ref2)
    assert c_ref["rays"] == c_sah["rays"]
    assert c_ref["box_tests"] > 2 * c_sah["box_tests"]          # median split on a random x/y axis culls far worse
    assert f_ref.info()["n_nodes"] >= f_sah.info()["n_nodes"]    # one object per leaf, spans of one stored twice


def test_binary_ply_matches_ascii(rtsr, orc, tmp_path):
    import struct
    verts = [(0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 1, 1)]
    faces = [(0, 1, 2), (0, 1, 3), (0, 2, 3), (1, 2, 3), (1, 2, 4)]
    asc = tmp_path / "a.ply"
    asc.write_text("ply\nformat ascii 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n"
                   "element face %d\nproperty list uchar int vertex_indices\nend_header\n" % (len(verts), len(faces)) +
                   "".join("%g %g %g\n" % v for v in verts) + "".join("3 %d %d %d\n" % f for f in faces))
    binp = tmp_path / "b.ply"
    header = ("ply\nformat binary_little_endian 1.0\ncomment extra vertex properties are skipped\nelement vertex %d\n"
              "property float x\nproperty float y\nproperty float z\nproperty uchar red\nproperty float confidence\n"
              "element face %d\nproperty list uchar int vertex_indices\nend_header\n" % (len(verts), len(faces))).encode()
    body = b"".join(struct.pack("<fffBf", *map(float, v), 200, 0.5) for v in verts)
    body += b"".join(struct.pack("<Biii", 3, *f) for f in faces)
    binp.write_bytes(header + body)
    cam = rtsr.Camera.new((3, 2, 4), (0.3, 0.3, 0.3), (0, 1, 0), 30.0, 1.0, 0.0, 5.0, 0.0, 1.0)
    cfg = rtsr.Config.new(1.0, 24, 2, 8, 1)
    imgs = []
    for path in (asc, binp):
        b = rtsr.Builder(1)
        world = b.bvh_from_list(b.triangle_model(str(path), 1.0), 0.0, 1.0)
        flat = b.flatten(world)
        assert flat.info()["n_triangles"] == len(faces)
        imgs.append(orc.o2_render(flat.arrays_ptr(), cam, cfg, 24)[0])
    assert np.array_equal(imgs[0], imgs[1])
    b = rtsr.Builder(1)
    bad = tmp_path / "c.ply"
    bad.write_bytes(header + body[:-5])
    with pytest.raises(rtsr.RtxError):
        b.triangle_model(str(bad), 1.0)


def test_time_aware_boxes_contain_their_spheres_and_cannot_be_seen(rtsr, orc):
    """FlatMotion32 (core/flat_types.hpp): for BVHs that hold MovingSpheres every child box, evaluated in f32 exactly as
    k_trace_lds evaluates it, contains every sphere below it at 24 instants of the BVH's interval; the CPU restatement of
    k_trace_lds's walk over those boxes renders the frame of the plain oracle bit for bit and visits far fewer nodes than the
    same walk over the reference's boxes (unions over the whole interval, hit.rs:317-327)."""
    for sid, expect_motion in ((13, True), (7, True), (100, False), (6, False)):
        b = rtsr.Builder(1)
        world, cam, bg = b.get_world_cam(sid)
        flat = b.flatten(world)
        rc, n = orc.audit_motion(flat.arrays_ptr(), 24)
        assert rc == 0, (sid, rc)
        assert (n > 0) == expect_motion, (sid, n)
    b = rtsr.Builder(1)
    world, cam, bg = b.get_world_cam(13)   # Book-1 as gen_random_scene builds it at HEAD: 80 % of the small spheres rise 5 units
    cfg = rtsr.Config.new(16.0 / 9.0, 96, 4, 50, 4, seed=3, background=bg)
    h = rtsr.image_height(cfg)
    flat = b.flatten(world)
    ref, _ = orc.o2_render(flat.arrays_ptr(), cam, cfg, h, threads=8)
    static, c_static = orc.lds_walk_render(flat.arrays_ptr(), cam, cfg, h, use_motion=False, threads=8)
    moving, c_moving = orc.lds_walk_render(flat.arrays_ptr(), cam, cfg, h, use_motion=True, threads=8)
    assert np.array_equal(static, ref) and np.array_equal(moving, ref)
    assert c_static["rays"] == c_moving["rays"]
    assert c_moving["node_visits"] < 0.65 * c_static["node_visits"], (c_static, c_moving)
    assert c_moving["prim_tests"] < 0.5 * c_static["prim_tests"], (c_static, c_moving)
