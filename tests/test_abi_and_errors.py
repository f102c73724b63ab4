"""The drop-in boundary: every symbol include/rtx_abi.h declares is exported by the built library and
bound by the Python layer; error behaviour mirrors the reference's asserts/panics as status codes.
No compute happens here (no GPU needed)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "rtx_abi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rtx_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(rtsr):
    declared = _declared_functions()
    assert len(declared) >= 45
    lib = C.CDLL(rtsr.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), "library does not export %s" % name
        assert name in rtsr.ABI, "python binding table lacks %s" % name
    assert sorted(rtsr.ABI) == declared
    assert rtsr.lib.rtx_abi_version() == 1


def test_library_is_in_tree_and_links_hip(rtsr):
    assert rtsr.LIB_PATH.startswith(ROOT) and os.path.exists(rtsr.LIB_PATH)
    blob = open(rtsr.LIB_PATH, "rb").read()
    assert b"amdgcn-amd-amdhsa--gfx950" in blob  # carries a gfx950 code object
    assert b"k_trace_persistent" in blob


def test_config_new_asserts(rtsr):  # world.rs:36-40
    for args in [(1.5, 0, 1, 1, 1), (1.5, 10, 0, 1, 1), (1.5, 10, 1, 0, 1), (1.5, 10, 1, 1, 0), (1.5, -5, 1, 1, 1)]:
        with pytest.raises(rtsr.RtxError) as e:
            rtsr.Config.new(*args)
        assert e.value.status == rtsr.RTX_EINVAL and "assert" in str(e.value)
    cfg = rtsr.Config.new(1.5, 10, 1, 1, 1)
    assert (cfg.seed, cfg.row_chunk_compat) == (1, 0) and tuple(cfg.background) == (0.7, 0.8, 1.0)


def test_camera_new(rtsr):  # camera.rs:20-57
    cam = rtsr.Camera.new((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, 1.5, 0.1, 10.0, 0.0, 1.0)
    assert tuple(cam.origin) == (13.0, 2.0, 3.0) and cam.lens_radius == 0.05
    w = tuple(cam.w)
    assert abs(sum(x * x for x in w) - 1.0) < 1e-15
    h_len = sum(x * x for x in cam.horizontal) ** 0.5
    v_len = sum(x * x for x in cam.vertical) ** 0.5
    assert abs(h_len / v_len - 1.5) < 1e-14  # viewport aspect
    with pytest.raises(rtsr.RtxError) as e:   # gen_range(time1..time2) panics on an empty range (camera.rs:69)
        rtsr.Camera.new((0, 0, 1), (0, 0, 0), (0, 1, 0), 20.0, 1.0, 0.0, 1.0, 1.0, 1.0)
    assert e.value.status == rtsr.RTX_EINVAL


def test_bad_handles_are_rejected(rtsr):
    b = rtsr.Builder(1)
    for call in (lambda: b.sphere((0, 0, 0), 1.0, 99), lambda: b.lambertian(12345), lambda: b.checker(0, 7),
                 lambda: b.translate((0, 0, 0), 50), lambda: b.bvh_from_list(3, 0, 1), lambda: b.rotate_y(10, -2)):
        with pytest.raises(rtsr.RtxError) as e:
            call()
        assert e.value.status == rtsr.RTX_EINVAL
    lst = b.hittable_list()
    with pytest.raises(rtsr.RtxError):   # BvhNode::new on an empty list panics in the reference
        b.bvh_from_list(lst, 0.0, 1.0)
    with pytest.raises(rtsr.RtxError):
        b.list_add(lst, lst)
    with pytest.raises(rtsr.RtxError) as e:
        b.flatten(777)
    assert e.value.status == rtsr.RTX_EINVAL


def test_unsupported_shapes_are_reported_not_approximated(rtsr):
    b = rtsr.Builder(1)
    m = b.lambertian((0.5, 0.5, 0.5))
    s = b.sphere((0, 0, 0), 1.0, m)
    deep = s
    for k in range(5):  # five wrappers: one more than RT_MAX_XFORM_OPS (chains of up to four flatten: tests/test_nesting.py)
        deep = b.translate((1, 0, 0), deep) if k % 2 else b.rotate_y(5.0, deep)
    with pytest.raises(rtsr.RtxError) as e:
        b.flatten(deep)
    assert e.value.status == rtsr.RTX_EUNSUPPORTED
    instanced = b.bvh_from_list(b.hittable_list([b.translate((1, 0, 0), s), s]), 0, 1)  # a wrapper INSIDE a BVH
    with pytest.raises(rtsr.RtxError) as e:
        b.flatten(instanced)
    assert e.value.status == rtsr.RTX_EUNSUPPORTED
    medium_in_bvh = b.bvh_from_list(b.hittable_list([b.constant_medium((1, 1, 1), 0.1, s), s]), 0, 1)
    with pytest.raises(rtsr.RtxError) as e:
        b.flatten(medium_in_bvh)
    assert e.value.status == rtsr.RTX_EUNSUPPORTED


def test_missing_files(rtsr, tmp_path):
    b = rtsr.Builder(1)
    with pytest.raises(rtsr.RtxError):
        b.image_from_ppm(str(tmp_path / "nope.ppm"))
    with pytest.raises(rtsr.RtxError):
        b.triangle_model(str(tmp_path / "nope.ply"), 1.0)
    assert "open" in rtsr.last_error().lower()


def _gpu_present():
    return os.path.exists("/dev/kfd")


@pytest.mark.skipif(_gpu_present(), reason="checks the no-GPU failure mode")
def test_no_cpu_fallback_without_gpu(rtsr):
    """The product must fail loudly when it cannot run on a GPU: there is no CPU render path."""
    b = rtsr.Builder(1)
    world, cam, bg = b.get_world_cam(rtsr.SCENE_TRIANGLE_TEST)
    flat = b.flatten(world)
    with pytest.raises(rtsr.RtxError) as e:
        flat.upload()
    assert e.value.status == rtsr.RTX_EHIP


def test_trace_kernel_names(rtsr):
    """RtxRenderStats.trace_kernel ids map to the kernel names rocprofv3 prints."""
    names = [rtsr.trace_kernel_name(k) for k in range(8)]
    assert names == ["k_trace_simple", "k_trace_persistent", "k_trace_stream", "k_trace_vote", "k_trace_lds", "k_trace_wq", "k_trace_world", "k_wf_trace"]
    assert rtsr.trace_kernel_name(99) == "?"
    for n in names:  # every reported name is a kernel that exists in the sources
        assert any(n in open(os.path.join(ROOT, "ray-tracing-series-rust_amd", "csrc", "hip", f)).read()
                   for f in os.listdir(os.path.join(ROOT, "ray-tracing-series-rust_amd", "csrc", "hip")) if f.endswith((".hip", ".inc")))


def test_product_never_touches_the_oracle():
    """oracle/ is test infrastructure: nothing under the product package may reference it."""
    pkg = os.path.join(ROOT, "ray-tracing-series-rust_amd")
    for base, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h", ".inc")):
                text = open(os.path.join(base, f), errors="replace").read()
                for line in text.splitlines():
                    code = line.split("//")[0].split("#")[0] if not f.endswith(".py") else line.split("#")[0]
                    assert "oracle_py" not in code and "liboracle" not in code and "oracle/" not in code.replace("oracle/ ", ""), (f, line)
