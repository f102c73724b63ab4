"""GPU parity for the reference-compat switches and the scene shapes the catalogue does not hold:
row_chunk_compat (world.rs:1198-1202), RtxBuildOptions.reference_bvh (bvh.rs:14-83), objects shared between
containers (the reference's Arc), zero-thickness boxes, and the PLY ingestion path (model.rs:13-76).

The two structural facts the reference's own images hold are pinned here on the HIP path:
  images/book2.png            1000x1000, 11 threads -> chunk 90, rows 990..999 never rendered = 10 black TOP rows
  images/stanford_dragon.png   600x375,  11 threads -> chunk 34, row 374 never rendered     =  1 black TOP row
(both were checked against the PNGs with PIL when the survey was written; the reference cannot travel to the GPU box).
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _top_rows_black(path):
    """Number of all-black rows at the top of a P3 PPM file written by the product."""
    with open(path) as f:
        assert f.readline().strip() == "P3"
        w, h = map(int, f.readline().split())
        assert f.readline().strip() == "255"
        px = np.loadtxt(f, dtype=np.int64).reshape(h, w, 3)  # first line = top row (screen.rs:43)
    n = 0
    while n < h and not px[n].any():
        n += 1
    return n, px


@pytest.mark.parametrize("name,sid,width,aspect,threads,black_top", [
    ("book2_png", 6, 1000, 1.0, 11, 10),           # default.cfg: threads=11, scene 6, 1000x1000
    ("stanford_dragon_png", 11, 600, 1.6, 11, 1),  # main.rs:4-11: THREADS=11, Config::new(1.6, 600, ...)
])
def test_reference_images_black_top_rows(rtsr, orc, tmp_path, name, sid, width, aspect, threads, black_top):
    b = rtsr.Builder(1)
    opts = {"mesh_triangles": 20000} if sid == 11 else {}
    world, cam, bg = b.get_world_cam(sid, **opts)
    cfg = rtsr.Config.new(aspect, width, 1, 50, threads, seed=2, background=bg, row_chunk_compat=True)
    h = rtsr.image_height(cfg)
    assert h == {6: 1000, 11: 375}[sid]
    flat = b.flatten(world)
    screen = flat.upload().render(cam, cfg)
    rendered = (h // threads) * threads
    assert h - rendered == black_top
    # rows j >= threads * floor(h / threads) stay (0,0,0) (Screen::new), every rendered row is lit somewhere
    assert not screen.accum[rendered:].any() and not screen.rgb8[rendered:].any()
    assert (screen.accum[:rendered].reshape(rendered, -1).max(axis=1) > 0).all()
    # through the PPM writer: the black rows are the TOP rows of the file, exactly `black_top` of them
    path = str(tmp_path / (name + ".ppm"))
    rtsr.Screen(width, h, screen.rgb8).write_to_ppm_file(path)
    n_black, px = _top_rows_black(path)
    assert n_black == black_top
    # and the rendered part equals the oracle's (spot rows: the frame is a megapixel)
    for j in (0, rendered // 2, rendered - 1):
        ref, ref8 = orc.o2_render(flat.arrays_ptr(), cam, cfg, h, shard=(j, h, 1), threads=threads)  # the oracle reads the band count from `threads` too
        assert np.array_equal(screen.accum[j], ref[0]) and np.array_equal(screen.rgb8[j], ref8[0]), j
        assert np.array_equal(px[h - 1 - j], ref8[0].astype(np.int64))


@pytest.mark.parametrize("threads,h_expect", [(4, 26), (40, 26), (1, 26)])
def test_row_chunk_compat_small(rtsr, orc, threads, h_expect):
    """threads > height gives chunk_size 0: nothing is rendered, the frame stays black (world.rs:1198)."""
    b = rtsr.Builder(1)
    world, cam, bg = b.get_world_cam(100)
    cfg = rtsr.Config.new(1.5, 40, 3, 50, threads, seed=5, background=bg, row_chunk_compat=True)
    h = rtsr.image_height(cfg)
    assert h == h_expect
    flat = b.flatten(world)
    screen = flat.upload().render(cam, cfg)
    a1, r1 = orc.o1_render(b.graph_ptr(), world, cam, cfg, h)
    assert np.array_equal(screen.accum, a1) and np.array_equal(screen.rgb8, r1)
    if threads > h:
        assert not screen.accum.any()
    # sharded: every shard leaves its own part of the skipped rows black
    out = np.full_like(a1, -1.0)
    scene = flat.upload()
    import torch
    for r in range(3):
        rows = [j for j in range(h) if j % 3 == r]
        d = torch.full((len(rows) * 40 * 3,), -1.0, dtype=torch.float64, device="cuda")
        scene.render_device(cam, cfg, shard=(r, 3, 1), d_accum=d.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        out[rows] = d.cpu().numpy().reshape(len(rows), 40, 3)
    assert np.array_equal(out, a1)


REFERENCE_BVH_CASES = [  # (name, scene id, width, aspect, spp, options, expected kernel)
    ("book1_canonical", 100, 120, 1.5, 6, {}, "k_trace_lds"),
    ("book1_head", 13, 120, 16.0 / 9.0, 6, {}, "k_trace_lds"),
    ("dragon_room", 11, 120, 16.0 / 9.0, 4, {"mesh_triangles": 20000}, "k_trace_vote"),
    ("book2_final", 6, 80, 1.0, 4, {}, "k_trace_world"),
]


@pytest.mark.parametrize("name,sid,width,aspect,spp,opts,kernel", REFERENCE_BVH_CASES, ids=[c[0] for c in REFERENCE_BVH_CASES])
def test_reference_rule_trees_on_the_device(rtsr, orc, name, sid, width, aspect, spp, opts, kernel):
    """Trees built by the reference's rule (random axis in {x,y}, stable sort, median split, span-1 nodes holding
    the same object twice; bvh.rs:14-83) walked by the fast kernels: the image must equal O1's, which builds the
    same kind of tree with another axis stream, and the default SAH build's."""
    b = rtsr.Builder(1)
    world, cam, bg = b.get_world_cam(sid, **opts)
    cfg = rtsr.Config.new(aspect, width, spp, 50, 10, seed=4, background=bg)
    h = rtsr.image_height(cfg)
    a1, r1 = orc.o1_render(b.graph_ptr(), world, cam, cfg, h, threads=16)
    for bvh_seed in (1, 2):
        flat = b.flatten(world, reference_bvh=True, bvh_seed=bvh_seed)
        scene = flat.upload()
        st = scene.render_device(cam, cfg, want_stats=True)
        if name != "book1_canonical" and name != "book1_head":
            assert rtsr.trace_kernel_name(st.trace_kernel) == kernel
        screen = scene.render(cam, cfg)
        assert np.array_equal(screen.accum, a1), "bvh_seed %d: %d pixels differ" % (bvh_seed, int((np.abs(screen.accum - a1).max(axis=2) > 0).sum()))
        assert np.array_equal(screen.rgb8, r1)


def test_shared_objects_on_the_device(rtsr, orc):
    from test_oracle_pairs import _shared_cam_cfg, shared_worlds
    cam, cfg, h = _shared_cam_cfg(rtsr)
    for name, b, world in shared_worlds(rtsr):
        a1, r1 = orc.o1_render(b.graph_ptr(), world, cam, cfg, h, threads=8)
        for kw in ({}, {"reference_bvh": True, "bvh_seed": 3}):
            screen = b.flatten(world, **kw).upload().render(cam, cfg)
            assert np.array_equal(screen.accum, a1), (name, kw)
            assert np.array_equal(screen.rgb8, r1), (name, kw)


@pytest.mark.parametrize("env", [{}, {"RTX_TRACE_KERNEL": "simple"}, {"RTX_TRACE_KERNEL": "persistent"}, {"RTX_WIDE": "1"}],
                         ids=["default", "simple", "persistent", "wide"])
def test_axis_aligned_triangles_on_the_device(rtsr, orc, monkeypatch, env):
    """Zero-thickness leaf boxes through every walker (f64 boxes: simple; f32 culling: vote / persistent; 4-wide)."""
    from test_oracle_pairs import axis_aligned_world
    cam = rtsr.Camera.new((0.5, 3.0, 9.0), (0.0, 0.3, 0.0), (0.0, 1.0, 0.0), 55.0, 1.5, 0.0, 9.0, 0.0, 1.0)
    cfg = rtsr.Config.new(1.5, 96, 4, 8, 4, seed=3, background=(0.6, 0.7, 0.9))
    h = rtsr.image_height(cfg)
    b0, list_world = axis_aligned_world(rtsr, as_bvh=False)
    expect, expect8 = orc.o1_render(b0.graph_ptr(), list_world, cam, cfg, h, threads=8)
    b1, bvh_world = axis_aligned_world(rtsr, as_bvh=True)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    for kw in ({}, {"max_leaf": 1}, {"reference_bvh": True, "bvh_seed": 2}):
        scene = b1.flatten(bvh_world, **kw).upload()
        screen = scene.render(cam, cfg)
        assert np.array_equal(screen.accum, expect), kw
        assert np.array_equal(screen.rgb8, expect8), kw
        st = scene.render_count(cam, cfg)  # the counting run traces the same paths
        assert st.samples == 96 * h * 4


def _write_ply(path, verts, faces, binary):
    with open(path, "wb") as f:
        fmt = "binary_little_endian" if binary else "ascii"
        f.write(("ply\nformat %s 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n"
                 "element face %d\nproperty list uchar int vertex_indices\nend_header\n" % (fmt, len(verts), len(faces))).encode())
        if binary:
            f.write(np.asarray(verts, dtype="<f4").tobytes())
            for fc in faces:
                f.write(bytes([3]) + np.asarray(fc, dtype="<i4").tobytes())
        else:
            for v in verts:
                f.write(("%.9g %.9g %.9g\n" % tuple(v)).encode())
            for fc in faces:
                f.write(("3 %d %d %d\n" % tuple(fc)).encode())


@pytest.mark.parametrize("binary", [False, True], ids=["ascii", "binary"])
def test_ply_model_through_the_device(rtsr, orc, tmp_path, binary):
    """TriangleModel::load_from_file(path, scale).to_hittable() -> BvhNode (world.rs:684-687) on the HIP path."""
    n = 24
    verts, faces = [], []
    for i in range(n + 1):
        for j in range(n + 1):
            x, z = i / n * 2 - 1, j / n * 2 - 1
            verts.append((np.float32(x), np.float32(0.3 * np.sin(3 * x) * np.cos(2 * z)), np.float32(z)))
    for i in range(n):
        for j in range(n):
            a, bb, c, d = i * (n + 1) + j, (i + 1) * (n + 1) + j, (i + 1) * (n + 1) + j + 1, i * (n + 1) + j + 1
            faces += [(a, bb, c), (a, c, d)]
    path = str(tmp_path / "wave.ply")
    _write_ply(path, verts, faces, binary)
    b = rtsr.Builder(1)
    model = b.triangle_model(path, 3.0)
    light = b.xz_rect(-2, 2, -2, 2, 4.0, b.diffuse_light((4.0, 4.0, 4.0)))
    world = b.hittable_list([b.bvh_from_list(model, 0.0, 1.0), light])
    cam = rtsr.Camera.new((0.0, 4.0, 7.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 50.0, 1.5, 0.0, 7.0, 0.0, 1.0)
    cfg = rtsr.Config.new(1.5, 96, 4, 12, 4, seed=8, background=(0.5, 0.6, 0.8))
    h = rtsr.image_height(cfg)
    flat = b.flatten(world)
    assert flat.info()["n_triangles"] == 2 * n * n
    screen = flat.upload().render(cam, cfg)
    a1, r1 = orc.o1_render(b.graph_ptr(), world, cam, cfg, h, threads=8)
    assert np.array_equal(screen.accum, a1) and np.array_equal(screen.rgb8, r1)
    assert screen.accum.std() > 0.01


def test_ppm_image_texture_on_the_device(rtsr, orc, tmp_path):
    """Image::from_ppm (texture.rs:95-99 -> Screen::from_ppm_p3, screen.rs:61-95) feeding an Image texture on the HIP path: a P3
    file written here, read by rtx_image_from_ppm, wrapped on a sphere and on a rectangle (sphere uv from the outward normal,
    hit.rs:226-236; rectangle uv, hit.rs:486-489), rendered by the device and by the literal oracle O1."""
    w, h = 64, 32
    yy, xx = np.mgrid[0:h, 0:w]
    rgb = np.stack([(xx * 4) % 256, (yy * 8) % 256, ((xx // 8 + yy // 8) % 2) * 255], axis=2).astype(np.uint8)
    path = tmp_path / "tex.ppm"
    path.write_text("P3\n%d %d\n255\n" % (w, h) + "".join("%d %d %d\n" % tuple(p) for row in rgb for p in row))
    b = rtsr.Builder(1)
    tex = b.image_from_ppm(str(path))
    world = b.hittable_list([b.sphere((0.0, 0.0, 0.0), 2.0, b.lambertian(tex)),
                             b.xy_rect(-6.0, 6.0, -3.0, 3.0, -2.5, b.lambertian(tex)),
                             b.sphere((0.0, 6.0, 4.0), 1.5, b.diffuse_light((6.0, 6.0, 6.0)))])
    cam = rtsr.Camera.new((3.0, 1.0, 9.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 40.0, 1.5, 0.0, 9.0, 0.0, 1.0)
    cfg = rtsr.Config.new(1.5, 192, 16, 12, 4, seed=4, background=(0.3, 0.3, 0.4))
    hgt = rtsr.image_height(cfg)
    flat = b.flatten(world)
    assert flat.info()["n_texels"] == w * h and flat.info()["n_images"] == 1
    screen = flat.upload().render(cam, cfg)
    a1, r1 = orc.o1_render(b.graph_ptr(), world, cam, cfg, hgt, threads=16)
    assert np.array_equal(screen.accum, a1) and np.array_equal(screen.rgb8, r1)
    assert len(np.unique(screen.rgb8.reshape(-1, 3), axis=0)) > 500  # the picture is on the surfaces


def _movers_world(rtsr):
    """Spheres that move along ALL three axes (Book-1 at HEAD moves along y only), static ones among them, one BVH over (0, 1)."""
    b = rtsr.Builder(5)
    mats = [b.lambertian((0.7, 0.3, 0.3)), b.metal((0.8, 0.8, 0.7), 0.1), b.dielectric(1.5), b.lambertian((0.3, 0.6, 0.8))]
    objs = [b.sphere((0.0, -300.0, 0.0), 300.0, b.lambertian(b.checker_from_colors((0.2, 0.3, 0.1), (0.9, 0.9, 0.9))))]
    for k in range(150):
        c0 = (8.0 * b.random() - 4.0, 0.25 + 1.5 * b.random(), 8.0 * b.random() - 4.0)
        r = 0.12 + 0.2 * b.random()
        if k % 4 == 0:
            objs.append(b.sphere(c0, r, mats[k % 4]))
        else:
            c1 = (c0[0] + 3.0 * b.random() - 1.5, c0[1] + 2.0 * b.random() - 0.5, c0[2] + 3.0 * b.random() - 1.5)
            objs.append(b.moving_sphere(c0, c1, 0.0, 1.0, r, mats[k % 4]))
    return b, b.hittable_list([b.bvh_from_list(b.hittable_list(objs), 0.0, 1.0)])


@pytest.mark.parametrize("shutter,expect_motion", [((0.0, 1.0), True), ((0.25, 0.6), True), ((0.5, 1.5), False), ((-1.0, 0.5), False)])
def test_time_aware_boxes_on_the_device(rtsr, orc, monkeypatch, shutter, expect_motion):
    """k_trace_lds on FlatMotion32 boxes (spheres moving along all axes) against the CPU oracle: bit-identical while the shutter
    lies inside the BVH's interval; outside it the launcher goes back to the reference's boxes (the lerp is no bound there)."""
    b, world = _movers_world(rtsr)
    cam = rtsr.Camera.new((7.0, 3.0, 8.0), (0.0, 0.8, 0.0), (0.0, 1.0, 0.0), 35.0, 1.5, 0.05, 10.0, shutter[0], shutter[1])
    cfg = rtsr.Config.new(1.5, 160, 12, 30, 4, seed=9, background=(0.6, 0.7, 0.9))
    h = rtsr.image_height(cfg)
    flat = b.flatten(world)
    assert orc.audit_motion(flat.arrays_ptr(), 32)[0] == 0
    ref, ref8 = orc.o2_render(flat.arrays_ptr(), cam, cfg, h, threads=16)
    if expect_motion:  # inside the interval the literal object graph agrees too (outside, its own random tree decides what it culls)
        a1, _ = orc.o1_render(b.graph_ptr(), world, cam, cfg, h, threads=16)
        assert np.array_equal(a1, ref)
        walk, c_motion = orc.lds_walk_render(flat.arrays_ptr(), cam, cfg, h, use_motion=True, threads=16)
        _, c_static = orc.lds_walk_render(flat.arrays_ptr(), cam, cfg, h, use_motion=False, threads=16)
        assert np.array_equal(walk, ref) and c_motion["node_visits"] < c_static["node_visits"]
    scene = flat.upload()
    st = scene.render_device(cam, cfg, want_stats=True)
    assert rtsr.trace_kernel_name(st.trace_kernel) == "k_trace_lds"
    screen = scene.render(cam, cfg)
    monkeypatch.setenv("RTX_MOTION", "0")
    plain = flat.upload().render(cam, cfg)
    assert np.array_equal(plain.accum, screen.accum) and np.array_equal(plain.rgb8, screen.rgb8)
    if expect_motion:
        assert np.array_equal(screen.accum, ref) and np.array_equal(screen.rgb8, ref8)
    else:
        # Ray times beyond the interval: a sphere may be OUTSIDE the box the reference gives it (hit.rs:317-327 bounds it by its
        # boxes at time0 and time1 only), and whether it is still found depends on which boxes a walk happens to enter before
        # it has a closer hit -- in the reference on its random tree, here on the walker.  No parity claim exists out there;
        # what is checked is that the launcher fell back to those boxes (same frame as with RTX_MOTION=0, above) and renders.
        assert np.isfinite(screen.accum).all() and screen.accum.std() > 0.1
