"""The scene catalogue (csrc/host/scenes.cpp) against constants derived BY HAND from the reference's builders.

The catalogue is shared by the product and by both CPU oracles, so a mis-transcribed scene would pass every parity
test.  These numbers were counted from /root/reference/src/world.rs, not from this build:

  final_scene (world.rs:494-616)       list.add x 12: BvhNode(400 RectPrism), XzRect light, MovingSphere, Sphere x3,
                                       ConstantMedium, Sphere r=5000, ConstantMedium, Sphere (earth), Sphere (marble),
                                       Translate(RotateY(BvhNode(1000 Sphere)))
  stanford_dragon (world.rs:681-751)   list.add x 8: BvhNode(model), XyRect x2, XzRect x2, YzRect x2, XzRect light
  gen_random_scene (world.rs:95-167)   root IS the BvhNode (world.rs:162-166): ground + <= 22*22 small + 3 big spheres;
                                       a small sphere is a MovingSphere iff choose_mat < 0.8 (world.rs:128)
  cameras / backgrounds                world.rs:1009-1029 (6), 1114-1134 (11), 1157-1177 (default), 879
"""
import math

import numpy as np

PRIM, GROUP, BVH, XFORM, MEDIUM = 0, 1, 2, 3, 4


def _cam_fields(cam):
    f = lambda n: np.array(list(getattr(cam, n)))
    return f("origin"), f("horizontal"), f("vertical"), cam.lens_radius, cam.time1, cam.time2


def test_book2_final_scene_shape(rtsr):
    b = rtsr.Builder(1)
    world, cam, bg = b.get_world_cam(rtsr.SCENE_BOOK2_FINAL)
    flat = b.flatten(world)
    info = flat.info()
    assert flat.top_level_kinds() == [BVH, PRIM, PRIM, PRIM, PRIM, PRIM, MEDIUM, PRIM, MEDIUM, PRIM, PRIM, XFORM]
    assert info["n_top_level"] == 12
    assert info["n_rects"] == 400 * 6 + 1            # 20 x 20 RectPrisms of six rectangles + the light
    assert info["n_moving_spheres"] == 1
    # 1000 instanced + glass, fuzzy metal, blue-glass shell and its medium boundary, r=5000 glass and its medium
    # boundary (the reference adds that sphere twice: as an object and as the fog's boundary, world.rs:564-577),
    # earth, marble
    assert info["n_spheres"] == 1000 + 8
    assert info["n_triangles"] == 0
    assert info["n_bvh"] == 2
    assert info["n_perlins"] == 1 and info["n_images"] == 1
    origin, horiz, vert, lens, t1, t2 = _cam_fields(cam)
    assert origin.tolist() == [478.0, 278.0, -600.0] and lens == 0.0 and (t1, t2) == (0.0, 1.0)
    # vfov 40, aspect 1, focus 10: |vertical| = |horizontal| = 2 tan(20 deg) * 10
    assert abs(np.linalg.norm(vert) - 20.0 * math.tan(math.radians(20.0))) < 1e-12
    assert abs(np.linalg.norm(horiz) - np.linalg.norm(vert)) < 1e-12
    assert tuple(bg) == (0.0, 0.0, 0.0)
    # options scale the two loops and nothing else
    b2 = rtsr.Builder(1)
    w2, _, _ = b2.get_world_cam(rtsr.SCENE_BOOK2_FINAL, book2_boxes_per_side=3, book2_spheres=10)
    i2 = b2.flatten(w2).info()
    assert i2["n_rects"] == 9 * 6 + 1 and i2["n_spheres"] == 10 + 8 and i2["n_top_level"] == 12


def test_dragon_room_shape(rtsr):
    b = rtsr.Builder(1)
    world, cam, bg = b.get_world_cam(rtsr.SCENE_STANFORD_DRAGON, mesh_triangles=5000)
    flat = b.flatten(world)
    info = flat.info()
    assert flat.top_level_kinds() == [BVH] + [PRIM] * 7
    assert info["n_rects"] == 7 and info["n_spheres"] == 0 and info["n_moving_spheres"] == 0
    assert 4000 <= info["n_triangles"] <= 6000 and info["n_bvh"] == 1
    # model.rs:72 makes one Lambertian per face; the build shares one (same albedo): 1 mesh + 6 walls/light... counted:
    # backdrop, backwall, ground, ceiling, left, right, light = 7 materials + the mesh's
    assert info["n_materials"] == 8
    origin, horiz, vert, lens, t1, t2 = _cam_fields(cam)
    assert origin.tolist() == [0.0, 20.0, 20.0] and lens == 0.0 and (t1, t2) == (0.0, 10.0)
    assert abs(np.linalg.norm(vert) - 2.0 * math.tan(math.radians(30.0)) * 40.0) < 1e-9          # vfov 60, focus 40
    assert abs(np.linalg.norm(horiz) / np.linalg.norm(vert) - 16.0 / 9.0) < 1e-12                # camera aspect 16/9 (Q2)
    assert tuple(bg) == (0.7, 0.8, 1.0)


def test_book1_scene_shapes(rtsr):
    for sid, moving in ((rtsr.SCENE_BOOK1_HEAD, True), (rtsr.SCENE_BOOK1_CANONICAL, False)):
        counts = []
        for seed in (1, 2, 3):
            b = rtsr.Builder(seed)
            world, cam, bg = b.get_world_cam(sid)
            flat = b.flatten(world)
            info = flat.info()
            assert flat.top_level_kinds() == [BVH]                      # the root is the BvhNode itself
            n = info["n_spheres"] + info["n_moving_spheres"]
            assert 484 - 12 <= n - 4 <= 484                               # a few candidates fall near (4, 0.2, 0)
            assert info["n_rects"] == 0 and info["n_triangles"] == 0 and info["n_bvh"] == 1
            if moving:
                frac = info["n_moving_spheres"] / float(n - 4)
                assert 0.7 < frac < 0.9                                   # choose_mat < 0.8 -> MovingSphere
            else:
                assert info["n_moving_spheres"] == 0
            counts.append(n)
            origin, horiz, vert, lens, t1, t2 = _cam_fields(cam)
            assert origin.tolist() == [13.0, 2.0, 3.0] and lens == 0.05   # aperture 0.1 / 2 (camera.rs:52)
            assert (t1, t2) == ((0.0, 10.0) if moving else (0.0, 1.0))
            assert abs(np.linalg.norm(vert) - 2.0 * math.tan(math.radians(10.0)) * 10.0) < 1e-12   # vfov 20, focus 10
            assert tuple(bg) == (0.7, 0.8, 1.0)
        assert len(set(counts)) > 1 or counts[0] != 488                   # layouts depend on the scene seed


def test_small_catalogue_scenes(rtsr):
    """cornell_box (world.rs:344-413): 6 walls/light + 2 Translate(RotateY(RectPrism)); cornell_smoke
    (world.rs:415-492): the two boxes become ConstantMedium boundaries; triangle_test (world.rs:665-679)."""
    b = rtsr.Builder(1)
    w, cam, bg = b.get_world_cam(rtsr.SCENE_CORNELL_BOX)
    f = b.flatten(w)
    assert f.top_level_kinds() == [PRIM] * 6 + [XFORM, XFORM] and f.info()["n_rects"] == 6 + 12
    assert tuple(bg) == (0.0, 0.0, 0.0)
    w, cam, bg = b.get_world_cam(rtsr.SCENE_CORNELL_SMOKE)
    f = b.flatten(w)
    assert f.top_level_kinds() == [PRIM] * 6 + [MEDIUM, MEDIUM] and f.info()["n_rects"] == 6 + 12
    w, cam, bg = b.get_world_cam(rtsr.SCENE_TRIANGLE_TEST)
    f = b.flatten(w)
    assert f.info()["n_triangles"] == 1 and f.info()["n_spheres"] == 1
