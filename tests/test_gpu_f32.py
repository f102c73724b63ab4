"""The f32 fast mode (rtx_scene_upload_f32) against the bit-exact f64 path on the GPU.

No bit-exactness is claimed for this mode; what is claimed -- and held here -- is that it renders the same picture:
the same RNG stream per (pixel, sample) and the same sample order, so images agree pixel by pixel up to the few paths
whose discrete decisions flip, and the frame's mean radiance agrees to a fraction of a percent.  The bars below sit
3-4x above what profiles/r02/f32_check.txt measured (mean within 1.4e-3 on every catalogue scene).

Three hazards that only exist with single-precision rays are pinned, each found as a failure while this mode was built:
  * a direction component of exactly 0 (1/d = inf) must neither break the child ordering of a 4-wide BVH step (GPU
    memory fault) nor switch culling off (one ray walking 871 200 triangles: 0.4 s);
  * a ray leaving a big sphere at a shallow angle must not find that sphere again (every sphere scene 0.2-1.2 % darker);
  * a non-finite ray ends its path instead of visiting every node.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CASES = [
    # name, scene id, aspect, width, spp, catalogue options
    ("book1", 100, 1.5, 240, 64, {}),
    ("book1_head", 13, 1.5, 240, 64, {}),
    ("cornell_box", 4, 1.0, 160, 128, {}),
    ("cornell_smoke", 5, 1.0, 160, 128, {}),
    ("book2_final", 6, 1.0, 200, 128, {}),
    ("two_perlin", 1, 1.5, 200, 64, {}),
    ("earth", 2, 1.5, 200, 64, {}),
    ("dragon_room", 11, 16.0 / 9.0, 320, 48, {"mesh_triangles": 60000}),
    ("random_moving", 8, 16.0 / 9.0, 240, 48, {}),
]


def _both(rtsr, sid, aspect, width, spp, opts):
    b = rtsr.Builder(1)
    world, cam, bg = b.get_world_cam(sid, **opts)
    flat = b.flatten(world)
    cfg = rtsr.Config.new(aspect, width, spp, 50, 10, seed=1, background=bg)
    out = {}
    for f32 in (False, True):
        scene = flat.upload(f32=f32)
        assert scene.is_f32 == f32
        stats = scene.render_device(cam, cfg, want_stats=True)      # timed by HIP events inside the library
        out[f32] = (scene.render(cam, cfg), stats)
    return flat, cam, cfg, out


@pytest.mark.parametrize("name,sid,aspect,width,spp,opts", CASES, ids=[c[0] for c in CASES])
def test_f32_renders_the_same_picture(rtsr, name, sid, aspect, width, spp, opts):
    _, _, _, out = _both(rtsr, sid, aspect, width, spp, opts)
    a = out[False][0].accum / spp
    b = out[True][0].accum / spp
    assert np.isfinite(b).all() or not np.isfinite(a).all()   # no NaN / inf pixels the f64 image does not have
    rel = abs(b.mean() - a.mean()) / a.mean()
    assert rel < 5e-3, (name, a.mean(), b.mean())
    la, lb = a.mean(axis=2), b.mean(axis=2)
    close = np.abs(la - lb) <= 0.05 * np.abs(la) + 0.02
    assert close.mean() > 0.90, (name, close.mean())
    # same kernel family as the f64 path, and no runaway walk: the fast mode is never slower than 1.5x the f64 kernel
    # (the zero-slope bug made the dragon room 22x slower; the shallow-angle bug made Book-2 1.25x slower)
    assert out[True][1].trace_kernel == out[False][1].trace_kernel
    assert out[True][1].trace_ms < 1.5 * out[False][1].trace_ms + 1.0, (name, out[True][1].trace_ms, out[False][1].trace_ms)


def test_f32_is_deterministic_and_shards_like_f64(rtsr):
    """Which lane, wave or GPU traces a sample is invisible in f32 too: two renders agree bit for bit, and a frame
    rendered as two row shards equals the unsharded frame."""
    import torch
    b = rtsr.Builder(1)
    world, cam, bg = b.get_world_cam(rtsr.SCENE_BOOK2_FINAL)
    flat = b.flatten(world)
    cfg = rtsr.Config.new(1.0, 96, 16, 50, 10, seed=3, background=bg)
    scene = flat.upload(f32=True)
    one = scene.render(cam, cfg)
    two = flat.upload(f32=True).render(cam, cfg)
    assert np.array_equal(one.accum, two.accum) and np.array_equal(one.rgb8, two.rgb8)
    h = rtsr.image_height(cfg)
    whole = np.zeros_like(one.accum)
    for idx in range(2):
        rows = rtsr.shard_rows(cfg, (idx, 2, 1))
        acc = torch.zeros((rows, 96, 3), dtype=torch.float64, device="cuda")
        scene.render_device(cam, cfg, shard=(idx, 2, 1), d_accum=acc.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        whole[idx::2][:rows] = acc.cpu().numpy()
    assert whole.shape[0] == h and np.array_equal(whole, one.accum)


def test_f32_on_several_shards(rtsr):
    """rtx_multi_create_f32: the frame cut into 3 row-interleaved shards (rehearsed on one GPU) equals the unsharded f32 frame."""
    b = rtsr.Builder(1)
    world, cam, bg = b.get_world_cam(rtsr.SCENE_BOOK1_CANONICAL)
    flat = b.flatten(world)
    cfg = rtsr.Config.new(1.5, 90, 8, 50, 10, seed=2, background=bg)
    whole = flat.upload(f32=True).render(cam, cfg)
    parts = rtsr.MultiScene(flat, 3, device_ids=[0, 0, 0], f32=True).render(cam, cfg)
    assert np.array_equal(parts.accum, whole.accum) and np.array_equal(parts.rgb8, whole.rgb8)
    assert not np.array_equal(whole.accum, flat.upload().render(cam, cfg).accum)   # and it really is the other arithmetic


def test_f32_handle_contract(rtsr):
    b = rtsr.Builder(1)
    world, cam, bg = b.get_world_cam(rtsr.SCENE_TRIANGLE_TEST)
    flat = b.flatten(world)
    cfg = rtsr.Config.new(1.5, 32, 2, 10, 4, seed=1, background=bg)
    scene = flat.upload(f32=True)
    assert scene.is_f32 and not flat.upload().is_f32
    with pytest.raises(rtsr.RtxError) as e:   # the work counters belong to the f64 path
        scene.render_count(cam, cfg)
    assert e.value.status == rtsr.RTX_EUNSUPPORTED
    first = scene.render(cam, cfg)
    scene.trim()                               # the workspace comes back on the next render
    assert np.array_equal(scene.render(cam, cfg).accum, first.accum)


def test_wide_step_sort_key(rtsr):
    """walk_node_step4 ranks its four children by wide_key(entry distance, t_min): clamp into [t_min, 3e38], NaN -> t_min
    (v_med3_f32 returns min3 of its operands when one is a NaN).  Without the clamp an entry distance of +inf ties with the
    +inf that marks a missed child and a stack slot stays unwritten -- the GPU memory fault this mode first died of."""
    x = np.array([np.nan, np.inf, -np.inf, 5.0, -1.0, 1e39, 0.001, 3.5e38], dtype=np.float64)
    y = np.full_like(x, 0.001)
    got = rtsr.device_math("wide_key", x, y)
    t = float(np.float32(0.001))
    big = float(np.float32(3.0e38))
    assert list(got) == [t, big, t, 5.0, t, big, t, big]
