"""GPU parity: the HIP path through the C ABI vs the CPU oracle, bit for bit.

Tolerance: 0.  All arithmetic on the path is f64 with identical operation order on both sides
(-ffp-contract=off, IEEE div/sqrt, shared deterministic libm), so accumulators must be EQUAL, which
is stronger than the north-star bound (per-pixel L2 < 1e-4).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# (scene id, image width, image aspect, spp, scene options)
CASES = [
    ("book1_canonical_C1", 100, 200, 3.0 / 2.0, 10, {}),   # BASELINE configs[0]: 200x133, 10 spp, depth 50
    ("book1_head", 13, 160, 16.0 / 9.0, 8, {}),
    ("checkered", 0, 96, 16.0 / 9.0, 8, {}),
    ("two_perlin", 1, 96, 16.0 / 9.0, 8, {}),
    ("earth", 2, 96, 16.0 / 9.0, 8, {}),
    ("simple_light", 3, 96, 16.0 / 9.0, 8, {}),
    ("cornell_box", 4, 80, 1.0, 8, {}),
    ("cornell_smoke", 5, 80, 1.0, 8, {}),
    ("book2_final", 6, 100, 1.0, 8, {}),
    ("moving_test", 7, 96, 16.0 / 9.0, 8, {}),
    ("benchmark_test", 9, 96, 16.0 / 9.0, 4, {}),
    ("triangle_test", 10, 96, 16.0 / 9.0, 8, {}),
    ("dragon_mesh_50k", 11, 120, 16.0 / 9.0, 4, {"mesh_triangles": 50000}),
    ("triangular_prism", 12, 80, 1.0, 8, {}),
    ("empty_world", 101, 32, 16.0 / 9.0, 2, {}),
]


def _setup(rtsr, sid, width, aspect, spp, opts, seed=3):
    b = rtsr.Builder(1)
    world, cam, bg = b.get_world_cam(sid, **opts)
    cfg = rtsr.Config.new(aspect, width, spp, 50, 10, seed=seed, background=bg)
    flat = b.flatten(world)
    return b, world, cam, cfg, flat


@pytest.mark.parametrize("name,sid,width,aspect,spp,opts", CASES, ids=[c[0] for c in CASES])
def test_gpu_equals_oracle(rtsr, orc, name, sid, width, aspect, spp, opts):
    b, world, cam, cfg, flat = _setup(rtsr, sid, width, aspect, spp, opts)
    h = rtsr.image_height(cfg)
    scene = flat.upload()
    screen = scene.render(cam, cfg)
    ref_accum, ref_rgb8 = orc.o2_render(flat.arrays_ptr(), cam, cfg, h, threads=16)
    diff = np.abs(screen.accum - ref_accum).max(axis=2)
    assert np.array_equal(screen.accum, ref_accum), "%d of %d pixels differ, max |d| = %g" % (
        int((diff > 0).sum()), diff.size, diff.max())
    assert np.array_equal(screen.rgb8, ref_rgb8)
    # north-star metric, for the record: per-pixel L2 on post-gamma [0,1] values
    assert float(np.sqrt(((screen.rgb8.astype(np.float64) - ref_rgb8) ** 2).sum(axis=2)).max()) / 255.0 < 1e-4


def test_gpu_equals_literal_oracle_book1(rtsr, orc):
    """GPU vs O1 (the literal object-graph restatement with the reference's own BVH rule)."""
    b, world, cam, cfg, flat = _setup(rtsr, 100, 120, 1.5, 6, {})
    h = rtsr.image_height(cfg)
    screen = flat.upload().render(cam, cfg)
    a1, r1 = orc.o1_render(b.graph_ptr(), world, cam, cfg, h, threads=16)
    assert np.array_equal(screen.accum, a1)
    assert np.array_equal(screen.rgb8, r1)


def test_scheduling_independence(rtsr, monkeypatch):
    """Persistent/regenerating kernel vs the plain grid-stride kernel, one pass vs many passes."""
    b, world, cam, cfg, flat = _setup(rtsr, 13, 128, 16.0 / 9.0, 16, {})
    base = flat.upload().render(cam, cfg)
    monkeypatch.setenv("RTX_TRACE_KERNEL", "simple")
    simple = flat.upload().render(cam, cfg)
    monkeypatch.delenv("RTX_TRACE_KERNEL")
    assert np.array_equal(base.accum, simple.accum)
    cfg2 = rtsr.RtxConfig.from_buffer_copy(cfg)
    cfg2.sample_buffer_bytes = 128 * 72 * 24 * 3  # 3 samples per pass -> 6 passes
    multi = flat.upload().render(cam, cfg2)
    assert np.array_equal(base.accum, multi.accum)


# Every trace-kernel variant the library can be switched to must produce the oracle's accumulators:
# which wave, lane or kernel traces a sample is a scheduling matter and must be invisible in the result.
VARIANTS = [
    {"RTX_RING": "0"},                      # voting kernel, regeneration per lane (no LDS ring of primary rays)
    {"RTX_RING": "1"},                      # voting kernel + ring (default where LDS allows)
    {"RTX_TRACE_KERNEL": "persistent"},
    {"RTX_TRACE_KERNEL": "simple"},
    {"RTX_WALK_THRESHOLD": "1", "RTX_LEAF_WEIGHT": "1"},
    {"RTX_LDS_WIDE": "1"},                  # k_trace_lds on the 4-wide collapse of the tree (measured slower: the A/B partner)
    {"RTX_LDS_WIDE": "1", "RTX_RING": "0"},
    {"RTX_MOTION": "0"},                    # HEAD Book-1 on the reference's boxes (unions over the shutter) instead of the time-aware ones
    {"RTX_MOTION_TOPOLOGY": "0"},           # ... and its tree partitioned by those boxes instead of the mid-interval ones
    {"RTX_LEAF_FIRST": "0"},                # plain near-child-by-split-axis order everywhere
    {"RTX_MOTION_AXIS": "0"},               # time-aware boxes with slopes for all three axes although only one moves (HEAD: y)
    {"RTX_SINGLE_LEAF": "0"},               # the general leaf loop on trees whose leaves all hold one primitive
    {"RTX_SINGLE_LEAF": "0", "RTX_RING": "0"},
]


@pytest.mark.parametrize("env", VARIANTS, ids=["-".join("%s=%s" % kv for kv in v.items()) for v in VARIANTS])
@pytest.mark.parametrize("sid,width,aspect,spp", [(100, 144, 1.5, 12), (13, 128, 16.0 / 9.0, 6)])
def test_kernel_variants_equal_oracle(rtsr, orc, monkeypatch, env, sid, width, aspect, spp):
    b, world, cam, cfg, flat = _setup(rtsr, sid, width, aspect, spp, {}, seed=11)
    h = rtsr.image_height(cfg)
    ref_accum, ref_rgb8 = orc.o2_render(flat.arrays_ptr(), cam, cfg, h, threads=16)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    screen = flat.upload().render(cam, cfg)
    assert np.array_equal(screen.accum, ref_accum)
    assert np.array_equal(screen.rgb8, ref_rgb8)


@pytest.mark.parametrize("sid", [100, 13])
@pytest.mark.parametrize("max_leaf", [2, 4])
def test_scene_in_lds_with_leaves_of_several_primitives(rtsr, orc, sid, max_leaf):
    """k_trace_lds keeps its sphere records in LDS in leaf-slot order and has a straight-line path for trees of one-primitive
    leaves (the default for spheres); here the same scenes with leaves of up to 2 / 4 spheres: the general leaf loop, slots
    f .. f + k - 1 of one leaf, static and moving spheres mixed in one leaf at HEAD."""
    b = rtsr.Builder(1)
    world, cam, bg = b.get_world_cam(sid)
    cfg = rtsr.Config.new(1.5, 132, 6, 50, 10, seed=21, background=bg)
    flat = b.flatten(world, max_leaf=max_leaf)
    h = rtsr.image_height(cfg)
    scene = flat.upload()
    st = scene.render_device(cam, cfg, want_stats=True)
    assert rtsr.trace_kernel_name(st.trace_kernel) == "k_trace_lds"
    screen = scene.render(cam, cfg)
    ref_accum, ref_rgb8 = orc.o2_render(flat.arrays_ptr(), cam, cfg, h, threads=16)
    assert np.array_equal(screen.accum, ref_accum)
    assert np.array_equal(screen.rgb8, ref_rgb8)


@pytest.mark.parametrize("scene_seed", [2, 3, 4, 5, 6])
@pytest.mark.parametrize("sid", [100, 13])
def test_other_scene_seeds(rtsr, orc, scene_seed, sid):
    """Different random sphere layouts give different tree depths and LDS footprints (with / without room for the
    primary-ray ring): the LDS kernel must agree with the oracle on all of them."""
    b = rtsr.Builder(scene_seed)
    world, cam, bg = b.get_world_cam(sid)
    cfg = rtsr.Config.new(1.5, 120, 6, 50, 10, seed=scene_seed, background=bg)
    flat = b.flatten(world)
    h = rtsr.image_height(cfg)
    scene = flat.upload()
    st = scene.render_device(cam, cfg, want_stats=True)
    assert rtsr.trace_kernel_name(st.trace_kernel) == "k_trace_lds"
    screen = scene.render(cam, cfg)
    ref_accum, ref_rgb8 = orc.o2_render(flat.arrays_ptr(), cam, cfg, h, threads=16)
    assert np.array_equal(screen.accum, ref_accum)
    assert np.array_equal(screen.rgb8, ref_rgb8)


@pytest.mark.parametrize("width,aspect,spp,depth", [(1, 1.0, 1, 1), (3, 1.5, 7, 2), (5, 0.5, 3, 50), (64, 1.5, 1, 1), (17, 1.5, 65, 3)])
def test_tiny_frames(rtsr, orc, width, aspect, spp, depth):
    """Frames smaller than a wave / a workgroup, one sample, depth 1: the persistent LDS kernel's start-up and
    drain logic (work chunks, the primary-ray ring, the end phase) on next to no work."""
    b = rtsr.Builder(1)
    world, cam, bg = b.get_world_cam(100, camera_aspect=aspect)
    cfg = rtsr.Config.new(aspect, width, spp, depth, 1, seed=21, background=bg)
    flat = b.flatten(world)
    h = rtsr.image_height(cfg)
    screen = flat.upload().render(cam, cfg)
    ref_accum, ref_rgb8 = orc.o2_render(flat.arrays_ptr(), cam, cfg, h, threads=2)
    assert np.array_equal(screen.accum, ref_accum)
    assert np.array_equal(screen.rgb8, ref_rgb8)


WIDE_CASES = [  # (name, scene id, width, aspect, spp, options, expected kernel)
    ("dragon_room", 11, 144, 16.0 / 9.0, 4, {"mesh_triangles": 20000}, "k_trace_vote"),
    ("book2_final", 6, 96, 1.0, 6, {}, "k_trace_world"),
    ("cornell_box", 4, 80, 1.0, 6, {}, "k_trace_world"),
    ("nested_lists", 9, 96, 16.0 / 9.0, 4, {}, "k_trace_world"),
]


@pytest.mark.parametrize("wide", ["0", "1"])
@pytest.mark.parametrize("name,sid,width,aspect,spp,opts,kernel", WIDE_CASES, ids=[c[0] for c in WIDE_CASES])
def test_binary_and_wide_culling_tree(rtsr, orc, monkeypatch, wide, name, sid, width, aspect, spp, opts, kernel):
    """Worlds walked through the binary and through the 4-wide culling tree (RTX_WIDE forces either)."""
    monkeypatch.setenv("RTX_WIDE", wide)
    b, world, cam, cfg, flat = _setup(rtsr, sid, width, aspect, spp, opts, seed=9)
    h = rtsr.image_height(cfg)
    scene = flat.upload()
    st = scene.render_device(cam, cfg, want_stats=True)
    assert rtsr.trace_kernel_name(st.trace_kernel) == kernel
    screen = scene.render(cam, cfg)
    ref_accum, ref_rgb8 = orc.o2_render(flat.arrays_ptr(), cam, cfg, h, threads=16)
    assert np.array_equal(screen.accum, ref_accum)
    assert np.array_equal(screen.rgb8, ref_rgb8)


def test_scenes_with_different_lds_footprints_coexist(rtsr, orc):
    """k_trace_lds sizes its dynamic LDS per scene; scenes uploaded earlier must keep rendering (and keep their
    results) after a scene with a different footprint has been uploaded and rendered."""
    setups = [_setup(rtsr, 100, 96, 1.5, 6, {}), _setup(rtsr, 7, 96, 16.0 / 9.0, 6, {}), _setup(rtsr, 13, 96, 16.0 / 9.0, 6, {})]
    scenes = [s[4].upload() for s in setups]
    first = [sc.render(s[2], s[3]).accum.copy() for sc, s in zip(scenes, setups)]
    for sc, s, f in reversed(list(zip(scenes, setups, first))):
        st = sc.render_device(s[2], s[3], want_stats=True)
        assert rtsr.trace_kernel_name(st.trace_kernel) == "k_trace_lds"
        assert np.array_equal(sc.render(s[2], s[3]).accum, f)
    b, world, cam, cfg, flat = setups[0]
    ref, _ = orc.o2_render(flat.arrays_ptr(), cam, cfg, rtsr.image_height(cfg), threads=16)
    assert np.array_equal(first[0], ref)


def test_list_ties_between_bvh_and_plain_entries(rtsr, orc):
    """HittableList::hit (hit.rs:660-690): on an exact tie the LATER list entry wins, wherever the BVH sits.

    World = [rect A, BVH{rect B coplanar with A, sphere}, rect C, rect D coplanar with C]: this shape takes the
    voting kernel's "one BVH + plain primitives" path, which tests the plain entries after the walk.  A/B tie must go
    to the BVH (later than A), C/D tie to D, and the GPU must agree with the literal oracle O1 on every pixel."""
    from test_oracle_pairs import tie_world
    b, world, cam, cfg, h = tie_world(rtsr)
    flat = b.flatten(world)
    scene = flat.upload()
    st = scene.render_device(cam, cfg, want_stats=True)
    assert rtsr.trace_kernel_name(st.trace_kernel) == "k_trace_vote"
    screen = scene.render(cam, cfg)
    a1, r1 = orc.o1_render(b.graph_ptr(), world, cam, cfg, h, threads=8)
    assert np.array_equal(screen.accum, a1)
    assert np.array_equal(screen.rgb8, r1)
    # the floor must come out green (the BVH's rect), never red, and the ceiling patch must emit
    floor_px = screen.accum[24, 48]  # row 0 is the bottom row; (24, 48) looks at the floor in front of the sphere
    assert floor_px[1] > 4.0 * floor_px[0]


def test_device_arithmetic_matches_host(rtsr, orc):
    rng = np.random.default_rng(5)
    n = 200000
    x = np.concatenate([rng.uniform(-1, 1, n), rng.uniform(-100, 100, n), rng.uniform(-6e4, 6e4, n)])
    for fn in ("sin", "cos", "tan"):
        assert np.array_equal(rtsr.device_math(fn, x), orc.rt_math(fn, x)), fn
    u = np.concatenate([rng.uniform(0, 1, n), rng.uniform(0, 1e-9, n), [0.0, 1.0, 2.0 ** -53]])
    assert np.array_equal(rtsr.device_math("log", u), orc.rt_math("log", u))
    a = np.concatenate([rng.uniform(-1, 1, n), [1.0, -1.0, 0.0, 1.0000000001]])
    assert np.array_equal(rtsr.device_math("acos", a), orc.rt_math("acos", a), equal_nan=True)
    yy, xx = rng.uniform(-1, 1, n), rng.uniform(-1, 1, n)
    assert np.array_equal(rtsr.device_math("atan2", yy, xx), orc.rt_math("atan2", yy, xx))
    p = np.abs(rng.normal(0, 1e3, n)) ** rng.uniform(0.1, 3, n)
    assert np.array_equal(rtsr.device_math("sqrt", p), np.sqrt(p))
    q = rng.normal(0, 10, n)
    d = rng.normal(0, 10, n)
    assert np.array_equal(rtsr.device_math("div", q, d), q / d)
    assert np.array_equal(rtsr.device_math("muladd", q, d), q * d + q)  # no FMA contraction
    assert np.array_equal(rtsr.device_math("floor", q), np.floor(q))


def test_device_stream_matches_host(rtsr, orc):
    import ctypes as C
    for seed, pixel, sample in [(1, 0, 0), (7, 123456, 499), (2 ** 40 + 5, 2 ** 33 + 1, 2 ** 31)]:
        host = np.empty(64)
        orc.load().oracle_sample_stream(seed, pixel, sample, 64, host.ctypes.data_as(C.POINTER(C.c_double)))
        assert np.array_equal(rtsr.device_stream(seed, pixel, sample, 64), host)
