"""GravitySphere (hit.rs:330-444), the bouncing-ball scene gen_random_scene_moving (world.rs:169-244, scene 8) and the
time-sweep renderer render_scene_with_time (world.rs:1249-1330).

Known answers derived by hand from GravitySphere::new / get_center:
  * stored[0] = start.y; the ball falls with vel -= 1e-6 per 0.001 time units, so after k steps y = y0 - 1e-6 k(k+1)/2
    (until the first bounce): at time 1.0 (k = 1000) y = y0 - 0.5005;
  * at rest it sits at y = radius (1.0 * radius in the table loop, hit.rs:353-356);
  * the table covers times [0, ~100.001): 100 001 or 100 002 entries (floating accumulation of t += 0.001);
  * past the table the OTHER loop runs (2 x radius, restitution 0.8; hit.rs:378-389): the ball then rests at 2 x radius.
"""
import numpy as np
import pytest


def _one_ball_world(rtsr, y0=3.0, radius=0.2):
    b = rtsr.Builder(1)
    ball = b.gravity_sphere((0.0, y0, 0.0), 0.0, radius, b.diffuse_light((10.0, 0.0, 0.0)))  # glows red: easy to find in the image
    ground = b.sphere((0.0, -1000.0, 0.0), 1000.0, b.lambertian((0.5, 0.5, 0.5)))
    return b, b.hittable_list([ground, ball])


def _cam(rtsr, t0, t1):
    return rtsr.Camera.new((0.0, 2.0, 9.0), (0.0, 1.5, 0.0), (0.0, 1.0, 0.0), 30.0, 1.0, 0.0, 9.0, t0, t1)


def _ball_top_row(accum):
    """Topmost image row (row 0 = bottom) on which the glowing ball shows in the centre column."""
    col = accum[:, accum.shape[1] // 2]
    rows = np.nonzero(col[:, 0] > 4.0 * col[:, 2] + 1.0)[0]
    return int(rows.max()) if rows.size else -1


def test_gravity_sphere_known_positions(rtsr, orc):
    """The ball is where the hand-derived trajectory puts it: rendered through a pinhole camera with a short shutter at
    several times, the ball's top edge must sit where y(t) + radius projects."""
    b, world = _one_ball_world(rtsr)
    flat = b.flatten(world)
    assert flat.info()["n_gravity_spheres"] == 1
    tops = []
    for t in (0.0, 1.0, 2.0):
        cam = _cam(rtsr, t, t + 0.001)
        cfg = rtsr.Config.new(1.0, 120, 4, 4, 4, seed=2, background=(0.1, 0.1, 0.4))
        a1, _ = orc.o1_render(b.graph_ptr(), world, cam, cfg, 120, threads=8)
        a2, _ = orc.o2_render(flat.arrays_ptr(), cam, cfg, 120, threads=8)
        assert np.array_equal(a1, a2)
        tops.append(_ball_top_row(a1))
    # y(0) = 3.0, y(1) = 3.0 - 0.5005, y(2) = 3.0 - 2.001: the ball moves down the image by the projected amounts
    assert tops[0] > tops[1] > tops[2] > 0
    # pinhole at (0,2,9) looking at (0,1.5,0), vfov 30: rows per world unit at the ball's depth ~ 120 / (2 * 9.01 * tan(15 deg))
    rows_per_unit = 120.0 / (2.0 * np.hypot(9.0, 0.5) * np.tan(np.radians(15.0)))
    assert abs((tops[0] - tops[1]) - 0.5005 * rows_per_unit) <= 2.0
    assert abs((tops[0] - tops[2]) - 2.001 * rows_per_unit) <= 2.5


def test_gravity_sphere_rest_and_fallback_loop(rtsr, orc):
    """Inside the table the ball comes to rest at y = radius; past the table (time > 100.002) the reference's second
    loop runs, which rests the ball at 2 x radius -- both through O1 (literal) and the flat path."""
    b, world = _one_ball_world(rtsr, y0=1.0, radius=0.25)
    flat = b.flatten(world)
    tops = {}
    for t in (99.9, 100.5):
        cam = _cam(rtsr, t, t + 0.0005)
        cfg = rtsr.Config.new(1.0, 96, 2, 3, 4, seed=4, background=(0.1, 0.1, 0.4))
        a1, _ = orc.o1_render(b.graph_ptr(), world, cam, cfg, 96, threads=8)
        a2, _ = orc.o2_render(flat.arrays_ptr(), cam, cfg, 96, threads=8)
        assert np.array_equal(a1, a2)
        tops[t] = _ball_top_row(a1)
    assert tops[100.5] > tops[99.9] > 0  # resting at 2 x radius shows higher than resting at radius


def test_scene_8_shape_and_oracle_pair(rtsr, orc):
    b = rtsr.Builder(1)
    world, cam, bg = b.get_world_cam(rtsr.SCENE_RANDOM_MOVING)
    flat = b.flatten(world)
    info = flat.info()
    # 22 x 22 grid minus the two 3 x 3 holes around (0,0) and (4,0) (world.rs:183-188) = 466 candidates, all of them
    # GravitySpheres (choose_mat < 1.0 always holds) unless within 0.9 of (4, 0.2, 0) -- impossible at y >= 1.7
    assert info["n_gravity_spheres"] == 22 * 22 - 18 and info["n_spheres"] == 4 and info["n_moving_spheres"] == 0
    assert info["n_bvh"] == 1 and info["n_top_level"] == 1
    assert (cam.time1, cam.time2, cam.lens_radius) == (0.0, 10.0, 0.05) and tuple(bg) == (0.7, 0.8, 1.0)
    cfg = rtsr.Config.new(16.0 / 9.0, 64, 2, 50, 4, seed=3, background=bg)
    h = rtsr.image_height(cfg)
    a1, r1 = orc.o1_render(b.graph_ptr(), world, cam, cfg, h, threads=8)
    a2, r2 = orc.o2_render(flat.arrays_ptr(), cam, cfg, h, threads=8)
    assert np.array_equal(a1, a2) and np.array_equal(r1, r2)


@pytest.mark.gpu
def test_scene_8_on_the_device(rtsr, orc):
    b = rtsr.Builder(1)
    world, cam, bg = b.get_world_cam(rtsr.SCENE_RANDOM_MOVING)
    flat = b.flatten(world)
    cfg = rtsr.Config.new(16.0 / 9.0, 128, 4, 50, 10, seed=3, background=bg)
    h = rtsr.image_height(cfg)
    scene = flat.upload()
    st = scene.render_device(cam, cfg, want_stats=True)
    assert rtsr.trace_kernel_name(st.trace_kernel) == "k_trace_world"
    screen = scene.render(cam, cfg)
    a1, r1 = orc.o1_render(b.graph_ptr(), world, cam, cfg, h, threads=16)
    assert np.array_equal(screen.accum, a1) and np.array_equal(screen.rgb8, r1)
    cnt = scene.render_count(cam, cfg)  # the counting kernel knows the primitive too
    assert cnt.samples == 128 * h * 4 and cnt.moving_sphere_tests > 0


@pytest.mark.gpu
def test_time_sweep_frames_on_a_resident_scene(rtsr, orc, tmp_path):
    """render_scene_with_time(t0, t1, path, world): several frames through ONE upload; each frame is the reference's
    hard-coded camera / config (world.rs:1252-1275) with the shutter [t0, t1), 11 row bands (5 black top rows of 500)."""
    b, world = _one_ball_world(rtsr)
    big = b.gravity_sphere((2.0, 2.5, 0.0), 0.0, 0.3, b.metal((0.8, 0.8, 0.9), 0.05))
    world = b.hittable_list([world, big])
    flat = b.flatten(world)
    scene = flat.upload()
    over = rtsr.Config.new(1.0, 100, 4, 20, 11, seed=9)  # keeps the test small: 100 x 100 x 4 spp
    frames = []
    for k, (t0, t1) in enumerate([(0.0, 0.1), (1.0, 1.1), (2.0, 2.1)]):
        path = str(tmp_path / ("frame%d.ppm" % k))
        scene.render_scene_with_time(t0, t1, path, row_chunk_compat=True, overrides=over)
        with open(path) as f:
            assert f.readline().strip() == "P3" and f.readline().split() == ["100", "100"] and f.readline().strip() == "255"
            px = np.loadtxt(f, dtype=np.int64).reshape(100, 100, 3)
        # 100 rows / 11 bands = 9 rows each: rows 99.. = 1 black TOP row of the file (the full-size frame: 500 / 11 -> 5)
        assert not px[0].any() and px[1].any()
        # the same frame through the generic entry point with the camera render_scene_with_time hard-codes
        cam = rtsr.Camera.new((13.0, 2.0, 3.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 20.0, 1.0, 0.1, 10.0, t0, t1)
        cfg = rtsr.Config.new(1.0, 100, 4, 20, 11, seed=9, background=(0.7, 0.8, 1.0), row_chunk_compat=True)
        ref = scene.render(cam, cfg)
        assert np.array_equal(px[::-1], ref.rgb8.astype(np.int64))  # file rows run top to bottom
        a1, r1 = orc.o1_render(b.graph_ptr(), world, cam, cfg, 100, threads=11)
        assert np.array_equal(ref.accum, a1)
        frames.append(px)
    assert not np.array_equal(frames[0], frames[2])  # the balls moved between frames
    with pytest.raises(rtsr.RtxError):  # t0 >= t1: gen_range(t0..t1) panics in the reference
        scene.render_scene_with_time(1.0, 1.0, str(tmp_path / "x.ppm"))


@pytest.mark.gpu
def test_time_sweep_frame_at_the_reference_size(rtsr, orc, tmp_path):
    """One frame of render_scene_with_time AS THE REFERENCE RENDERS IT (world.rs:1252-1275): 500 x 500, 500 spp, depth 50, 11 row
    bands, on scene 8 (466 GravitySpheres, world.rs:169-244) kept resident -- 125 million paths through rtx_render_scene_with_time
    with no overrides.  Checked: the PPM's shape, the 5 rows the band split drops, three of its rows against the literal oracle O1
    at the same 500 spp, and that the second frame of the sweep differs (the balls fell)."""
    b = rtsr.Builder(1)
    world, _, bg = b.get_world_cam(rtsr.SCENE_RANDOM_MOVING)
    flat = b.flatten(world)
    scene = flat.upload()
    t0, t1 = 0.2, 0.3
    path = str(tmp_path / "frame.ppm")
    scene.render_scene_with_time(t0, t1, path, row_chunk_compat=True)
    with open(path) as f:
        assert f.readline().strip() == "P3" and f.readline().split() == ["500", "500"] and f.readline().strip() == "255"
        px = np.loadtxt(f, dtype=np.int64).reshape(500, 500, 3)
    assert not px[:5].any() and px[5].any()  # 500 / 11 = 45 rows per band: rows 495..499 (the file's first five) are never rendered
    cam = rtsr.Camera.new((13.0, 2.0, 3.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 20.0, 1.0, 0.1, 10.0, t0, t1)
    cfg = rtsr.Config.new(1.0, 500, 500, 50, 11, seed=1, background=(0.7, 0.8, 1.0), row_chunk_compat=True)
    for top_row in (120, 300, 430):  # sky + balls, the big spheres, the ground
        acc = orc.o1_render_window(b.graph_ptr(), world, cam, cfg, 500, (top_row, top_row + 1, 0, 500), threads=16)
        want = np.array([orc.tone_map("o1", p, 500) for p in acc[0]])  # get_normalized_color, vec3.rs:89-107
        assert np.array_equal(px[top_row], want), top_row
    path2 = str(tmp_path / "frame2.ppm")
    over = rtsr.Config.new(1.0, 500, 8, 50, 11, seed=1)
    scene.render_scene_with_time(1.2, 1.3, path2, row_chunk_compat=True, overrides=over)
    scene.render_scene_with_time(t0, t1, path, row_chunk_compat=True, overrides=over)
    assert open(path2).read() != open(path).read()
