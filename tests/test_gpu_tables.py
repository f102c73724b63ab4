"""GPU parity of the table placements added in round 2: where a kernel reads a small read-only table from (LDS, the kernarg
segment, the material record) is invisible in the result.

  RTX_VOTE_TOP   k_trace_vote takes the plain entries beside the BVH (the dragon room's rectangles, hit.rs:476-631 as entries of
                 HittableList::hit, hit.rs:660-690) from its kernel arguments (default) / walks the world list (0)
  RTX_MAT_LDS    material + texture records in LDS behind the traversal stacks (k_trace_vote wide, k_trace_world) / in HBM (0)
  RTX_PERLIN_LDS Perlin tables (perlin.rs:6-11) in LDS (k_trace_world) / in HBM (0)
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CASES = [  # (name, scene id, width, aspect, spp, options, kernel)
    ("dragon_room", 11, 128, 16.0 / 9.0, 5, {"mesh_triangles": 20000}, "k_trace_vote"),
    ("book2_final", 6, 96, 1.0, 6, {}, "k_trace_world"),
    ("two_perlin", 1, 96, 16.0 / 9.0, 6, {}, "k_trace_world"),
    ("cornell_smoke", 5, 72, 1.0, 6, {}, "k_trace_world"),
]
ENVS = [{}, {"RTX_VOTE_TOP": "0"}, {"RTX_MAT_LDS": "0"}, {"RTX_PERLIN_LDS": "0"}, {"RTX_VOTE_TOP": "0", "RTX_MAT_LDS": "0", "RTX_PERLIN_LDS": "0"}]


@pytest.mark.parametrize("env", ENVS, ids=["default", "list_walk", "tables_in_hbm", "perlin_in_hbm", "all_off"])
@pytest.mark.parametrize("name,sid,width,aspect,spp,opts,kernel", CASES, ids=[c[0] for c in CASES])
def test_table_placement_is_invisible(rtsr, orc, monkeypatch, env, name, sid, width, aspect, spp, opts, kernel):
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    b = rtsr.Builder(1)
    world, cam, bg = b.get_world_cam(sid, **opts)
    cfg = rtsr.Config.new(aspect, width, spp, 50, 10, seed=29, background=bg)
    h = rtsr.image_height(cfg)
    flat = b.flatten(world)
    scene = flat.upload()
    st = scene.render_device(cam, cfg, want_stats=True)
    assert rtsr.trace_kernel_name(st.trace_kernel) == kernel
    screen = scene.render(cam, cfg)
    ref_accum, ref_rgb8 = orc.o2_render(flat.arrays_ptr(), cam, cfg, h, threads=16)
    assert np.array_equal(screen.accum, ref_accum)
    assert np.array_equal(screen.rgb8, ref_rgb8)


def test_room_with_a_sphere_beside_the_mesh_walks_the_list(rtsr, orc):
    """A plain entry that is not a rectangle keeps k_trace_vote on the world list (the kernarg table holds rectangle records only)."""
    b = rtsr.Builder(1)
    mesh_world, cam, bg = b.get_world_cam(11, mesh_triangles=5000)
    # the catalogue's room + one sphere appended to the world list
    red = b.lambertian((0.9, 0.2, 0.2))
    b.list_add(mesh_world, b.sphere((15.0, 12.0, 0.0), 4.0, red))
    cfg = rtsr.Config.new(1.6, 96, 4, 50, 10, seed=31, background=bg)
    h = rtsr.image_height(cfg)
    flat = b.flatten(mesh_world)
    scene = flat.upload()
    st = scene.render_device(cam, cfg, want_stats=True)
    assert rtsr.trace_kernel_name(st.trace_kernel) == "k_trace_vote"
    screen = scene.render(cam, cfg)
    ref_accum, ref_rgb8 = orc.o2_render(flat.arrays_ptr(), cam, cfg, h, threads=16)
    assert np.array_equal(screen.accum, ref_accum) and np.array_equal(screen.rgb8, ref_rgb8)
    a1, r1 = orc.o1_render(b.graph_ptr(), mesh_world, cam, cfg, h, threads=16)
    assert np.array_equal(screen.accum, a1)
