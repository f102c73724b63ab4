"""GPU parity of the wavefront (split-kernel) integrator, csrc/hip/trace_wave.inc: path state in HBM, one launch per
stage and bounce (k_wf_generate / k_wf_trace / k_wf_shade).  It replaces the same loop as every other trace kernel
(world.rs:1207-1226, ray_color world.rs:52-93, BvhNode::hit bvh.rs:97-112, HittableList::hit hit.rs:660-690), so the
accumulators must equal the CPU oracle's bit for bit -- whatever the pool size, the refill threshold, or how often the
host looks at the counters (all three only move work between launches).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CASES = [  # (name, scene id, width, aspect, spp, scene options)
    ("book1_final", 100, 96, 1.5, 9, {}),                                   # one BVH of static spheres: binary culling tree
    ("book1_head", 13, 96, 16.0 / 9.0, 5, {}),                              # moving spheres + checker ground: the ray's time travels
    ("dragon_room_wide", 11, 128, 16.0 / 9.0, 4, {"mesh_triangles": 20000}),  # mesh + 7 rectangles, 4-wide tree, tie ceiling / light
    ("dragon_room_binary", 11, 96, 16.0 / 9.0, 3, {"mesh_triangles": 600}),   # the same room, small mesh: binary tree
]
POOLS = [
    {},                                                  # default pool (larger than these frames: one generation)
    {"RTX_WF_PATHS": "512", "RTX_WF_CHECK": "1"},        # pool far smaller than the pass: slots are reused thousands of times
    {"RTX_WF_PATHS": "4096", "RTX_WF_REFILL": "1", "RTX_WF_CHECK": "3"},
    {"RTX_WF_PATHS": "1024", "RTX_WF_REFILL": "64", "RTX_WF_CHECK": "7"},
]


def _setup(rtsr, sid, width, aspect, spp, opts, seed):
    b = rtsr.Builder(1)
    world, cam, bg = b.get_world_cam(sid, **opts)
    cfg = rtsr.Config.new(aspect, width, spp, 50, 10, seed=seed, background=bg)
    return b, world, cam, cfg, b.flatten(world)


@pytest.mark.parametrize("pool", POOLS, ids=["default", "P512", "P4096-refill1", "P1024-refill64"])
@pytest.mark.parametrize("name,sid,width,aspect,spp,opts", CASES, ids=[c[0] for c in CASES])
def test_wavefront_equals_oracle(rtsr, orc, monkeypatch, pool, name, sid, width, aspect, spp, opts):
    b, world, cam, cfg, flat = _setup(rtsr, sid, width, aspect, spp, opts, seed=17)
    h = rtsr.image_height(cfg)
    ref_accum, ref_rgb8 = orc.o2_render(flat.arrays_ptr(), cam, cfg, h, threads=16)
    monkeypatch.setenv("RTX_TRACE_KERNEL", "wavefront")
    for k, v in pool.items():
        monkeypatch.setenv(k, v)
    scene = flat.upload()
    st = scene.render_device(cam, cfg, want_stats=True)
    assert rtsr.trace_kernel_name(st.trace_kernel) == "k_wf_trace"
    screen = scene.render(cam, cfg)
    assert np.array_equal(screen.accum, ref_accum)
    assert np.array_equal(screen.rgb8, ref_rgb8)


def test_wavefront_passes_shards_and_depth_one(rtsr, orc, monkeypatch):
    """Several passes over one pool, a sharded frame, and max_depth 1 (every path ends in its first shading)."""
    monkeypatch.setenv("RTX_TRACE_KERNEL", "wavefront")
    monkeypatch.setenv("RTX_WF_PATHS", "2048")
    b, world, cam, cfg, flat = _setup(rtsr, 11, 80, 16.0 / 9.0, 7, {"mesh_triangles": 5000}, seed=3)
    h = rtsr.image_height(cfg)
    scene = flat.upload()
    ref_accum, ref_rgb8 = orc.o2_render(flat.arrays_ptr(), cam, cfg, h, threads=16)
    cfg2 = rtsr.RtxConfig.from_buffer_copy(cfg)
    cfg2.sample_buffer_bytes = 80 * h * 24 * 2  # 2 samples per pass -> 4 passes
    screen = scene.render(cam, cfg2)
    assert np.array_equal(screen.accum, ref_accum) and np.array_equal(screen.rgb8, ref_rgb8)
    import torch
    for r in range(3):  # rows j with j % 3 == r
        rows = [j for j in range(h) if j % 3 == r]
        acc = torch.zeros((len(rows), 80, 3), dtype=torch.float64, device="cuda")
        rgb = torch.zeros((len(rows), 80, 3), dtype=torch.uint8, device="cuda")
        scene.render_device(cam, cfg, shard=(r, 3, 1), d_accum=acc.data_ptr(), d_rgb8=rgb.data_ptr())
        torch.cuda.synchronize()
        assert np.array_equal(acc.cpu().numpy(), ref_accum[rows]) and np.array_equal(rgb.cpu().numpy(), ref_rgb8[rows])
    cfg1 = rtsr.Config.new(16.0 / 9.0, 80, 5, 1, 10, seed=4, background=tuple(cfg.background))
    ref1, ref1_8 = orc.o2_render(flat.arrays_ptr(), cam, cfg1, h, threads=16)
    s1 = scene.render(cam, cfg1)
    assert np.array_equal(s1.accum, ref1) and np.array_equal(s1.rgb8, ref1_8)


def test_wavefront_does_not_apply_to_list_worlds(rtsr, monkeypatch):
    """Worlds with media / transforms / several BVHs keep k_trace_world even when the wavefront integrator is asked for."""
    monkeypatch.setenv("RTX_TRACE_KERNEL", "wavefront")
    b = rtsr.Builder(1)
    world, cam, bg = b.get_world_cam(4)  # Cornell box
    cfg = rtsr.Config.new(1.0, 32, 1, 5, 1, seed=1, background=bg)
    st = b.flatten(world).upload().render_device(cam, cfg, want_stats=True)
    assert rtsr.trace_kernel_name(st.trace_kernel) == "k_trace_world"
