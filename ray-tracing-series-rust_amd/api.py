"""ctypes binding of include/rtx_abi.h plus a host-side mirror of the reference's scene API.

The reference's host code is Rust; neither Rust nor a Rust binding can be built in this image,
so the compiled-language mirror lives in C++ (csrc/host, apps/rtx_render.cpp) and this module
is the Python face of the same C ABI: same names, same argument orders as the reference's
constructors (src/hit.rs, src/texture.rs, src/camera.rs, src/world.rs), so tests read like the
reference's own call sites.

There is no CPU render path here: if the HIP library is missing this module raises at import;
if no GPU is present, upload/render raise RtxError(RTX_EHIP).
"""
import ctypes as C
import os
import sys

import numpy as np

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
# RTX_LIBRARY: another build of the same library (scripts/ab_builds.sh: two builds timed in one session on one GPU)
_LIB_PATH = os.environ.get("RTX_LIBRARY") or os.path.join(_PKG_DIR, "lib", "librtx_hip.so")

RTX_OK, RTX_EINVAL, RTX_ENOMEM, RTX_EHIP, RTX_EUNSUPPORTED, RTX_EIO, RTX_ENCCL = 0, 1, 2, 3, 4, 5, 6
_STATUS_NAMES = {0: "RTX_OK", 1: "RTX_EINVAL", 2: "RTX_ENOMEM", 3: "RTX_EHIP", 4: "RTX_EUNSUPPORTED", 5: "RTX_EIO", 6: "RTX_ENCCL"}

SCENE_CHECKERED_SPHERES, SCENE_TWO_PERLIN, SCENE_EARTH, SCENE_SIMPLE_LIGHT = 0, 1, 2, 3
SCENE_CORNELL_BOX, SCENE_CORNELL_SMOKE, SCENE_BOOK2_FINAL, SCENE_MOVING_TEST = 4, 5, 6, 7
SCENE_RANDOM_MOVING, SCENE_BENCHMARK_TEST, SCENE_TRIANGLE_TEST, SCENE_STANFORD_DRAGON = 8, 9, 10, 11
SCENE_TRIANGULAR_PRISM, SCENE_BOOK1_HEAD, SCENE_BOOK1_CANONICAL, SCENE_EMPTY = 12, 13, 100, 101


class RtxError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("%s: %s" % (_STATUS_NAMES.get(status, status), message))
        self.status = status


class RtxCamera(C.Structure):
    _fields_ = [(n, C.c_double * 3) for n in ("origin", "lower_left_corner", "horizontal", "vertical", "u", "v", "w")] + \
               [("lens_radius", C.c_double), ("time1", C.c_double), ("time2", C.c_double)]


class RtxConfig(C.Structure):
    _fields_ = [("aspect_ratio", C.c_double), ("image_width", C.c_int32), ("samples_per_pixel", C.c_int32),
                ("max_depth", C.c_int32), ("threads", C.c_int32), ("seed", C.c_uint64),
                ("background", C.c_double * 3), ("row_chunk_compat", C.c_int32), ("reserved", C.c_int32),
                ("sample_buffer_bytes", C.c_uint64)]


class RtxSceneOptions(C.Structure):
    _fields_ = [("camera_aspect", C.c_double), ("earth_ppm", C.c_char_p), ("dragon_ply", C.c_char_p),
                ("mesh_triangles", C.c_int64), ("book2_boxes_per_side", C.c_int32), ("book2_spheres", C.c_int32)]


class RtxBuildOptions(C.Structure):
    _fields_ = [("max_leaf", C.c_int32), ("sah_bins", C.c_int32), ("reference_bvh", C.c_int32), ("gpu_builder", C.c_int32),
                ("bvh_seed", C.c_uint64)]


class RtxFlatInfo(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("n_spheres", "n_moving_spheres", "n_rects", "n_triangles", "n_nodes",
                                         "n_refs", "n_entries", "n_top_level", "n_materials", "n_textures",
                                         "n_perlins", "n_images", "n_texels", "total_bytes")] + \
               [("max_stack", C.c_int32), ("n_bvh", C.c_int32), ("sah_cost", C.c_double), ("bvh_build_ms", C.c_double),
                ("bvh_device_ms", C.c_double), ("n_gravity_spheres", C.c_int64)]


class RtxFrame(C.Structure):
    _fields_ = [("accum_rgb", C.POINTER(C.c_double)), ("rgb8", C.POINTER(C.c_uint8))]


class RtxRenderStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("samples", "rays", "box_tests", "sphere_tests", "moving_sphere_tests",
                                          "rect_tests", "triangle_tests", "scatters", "texels", "perlin_calls")] + \
               [("trace_ms", C.c_double), ("reduce_ms", C.c_double), ("tonemap_ms", C.c_double),
                ("trace_launches", C.c_int32), ("passes", C.c_int32), ("sample_buffer_bytes", C.c_uint64),
                ("trace_kernel", C.c_int32), ("reserved", C.c_int32)]


class RtxMultiStats(C.Structure):
    _fields_ = [("render_ms_max", C.c_double), ("total_ms", C.c_double), ("gathered_bytes", C.c_uint64),
                ("n_shards", C.c_int32), ("n_devices", C.c_int32), ("used_rccl", C.c_int32), ("rccl_ranks", C.c_int32),
                ("gather_ms", C.c_double), ("render_ms", C.c_double * 16)]


class RtxShard(C.Structure):
    _fields_ = [("shard_index", C.c_int32), ("shard_count", C.c_int32), ("block_rows", C.c_int32), ("reserved", C.c_int32)]


# Every symbol include/rtx_abi.h declares: (restype, argtypes).  tests/test_abi_symbols.py checks
# this table against the header and against the loaded library.
_D3 = C.POINTER(C.c_double)
_VP = C.c_void_p
_H = C.c_int32
ABI = {
    "rtx_abi_version": (C.c_int32, []),
    "rtx_last_error": (C.c_char_p, []),
    "rtx_trace_kernel_name": (C.c_char_p, [C.c_int32]),
    "rtx_builder_create": (C.c_int32, [C.c_uint64, C.POINTER(_VP)]),
    "rtx_builder_destroy": (None, [_VP]),
    "rtx_builder_random": (C.c_double, [_VP]),
    "rtx_solid_color": (_H, [_VP, _D3]),
    "rtx_checker": (_H, [_VP, _H, _H]),
    "rtx_noise": (_H, [_VP, C.c_double]),
    "rtx_image_from_ppm": (_H, [_VP, C.c_char_p]),
    "rtx_image_from_texels": (_H, [_VP, C.c_int32, C.c_int32, _D3]),
    "rtx_lambertian": (_H, [_VP, _H]),
    "rtx_metal": (_H, [_VP, _D3, C.c_double]),
    "rtx_dielectric": (_H, [_VP, C.c_double]),
    "rtx_diffuse_light": (_H, [_VP, _H]),
    "rtx_isotropic": (_H, [_VP, _H]),
    "rtx_sphere": (_H, [_VP, _D3, C.c_double, _H]),
    "rtx_moving_sphere": (_H, [_VP, _D3, _D3, C.c_double, C.c_double, C.c_double, _H]),
    "rtx_triangle": (_H, [_VP, _D3, _D3, _D3, _H]),
    "rtx_gravity_sphere": (_H, [_VP, _D3, C.c_double, C.c_double, _H]),
    "rtx_render_scene_with_time": (C.c_int32, [_VP, C.c_double, C.c_double, C.c_char_p, C.c_int32, C.POINTER(RtxConfig)]),
    "rtx_xy_rect": (_H, [_VP] + [C.c_double] * 5 + [_H]),
    "rtx_xz_rect": (_H, [_VP] + [C.c_double] * 5 + [_H]),
    "rtx_yz_rect": (_H, [_VP] + [C.c_double] * 5 + [_H]),
    "rtx_rect_prism": (_H, [_VP, _D3, _D3, _H]),
    "rtx_hittable_list_new": (_H, [_VP]),
    "rtx_hittable_list_add": (C.c_int32, [_VP, _H, _H]),
    "rtx_bvh_from_list": (_H, [_VP, _H, C.c_double, C.c_double]),
    "rtx_translate": (_H, [_VP, _D3, _H]),
    "rtx_rotate_y": (_H, [_VP, C.c_double, _H]),
    "rtx_constant_medium": (_H, [_VP, _D3, C.c_double, _H]),
    "rtx_triangle_model": (_H, [_VP, C.c_char_p, C.c_double]),
    "rtx_triangle_mesh": (_H, [_VP, _D3, C.c_int64, C.POINTER(C.c_int64), C.c_int64, _H]),
    "rtx_camera_new": (C.c_int32, [_D3, _D3, _D3] + [C.c_double] * 6 + [C.POINTER(RtxCamera)]),
    "rtx_config_new": (C.c_int32, [C.c_double, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(RtxConfig)]),
    "rtx_image_height": (C.c_int32, [C.POINTER(RtxConfig)]),
    "rtx_flat_top_level_kind": (C.c_int32, [_VP, C.c_int32]),
    "rtx_scene_trim": (C.c_int32, [_VP]),
    "rtx_multi_create": (C.c_int32, [_VP, C.c_int32, C.POINTER(C.c_int32), C.c_int32, C.POINTER(_VP)]),
    "rtx_multi_create_f32": (C.c_int32, [_VP, C.c_int32, C.POINTER(C.c_int32), C.c_int32, C.POINTER(_VP)]),
    "rtx_multi_destroy": (None, [_VP]),
    "rtx_multi_render": (C.c_int32, [_VP, C.POINTER(RtxCamera), C.POINTER(RtxConfig), C.POINTER(RtxFrame), C.POINTER(RtxMultiStats)]),
    "rtx_render_multi": (C.c_int32, [_VP, C.POINTER(RtxCamera), C.POINTER(RtxConfig), C.c_int32, C.POINTER(RtxFrame)]),
    "rtx_get_world_cam": (C.c_int32, [_VP, C.c_int32, C.POINTER(RtxSceneOptions), C.POINTER(_H), C.POINTER(RtxCamera), _D3]),
    "rtx_flatten": (C.c_int32, [_VP, _H, C.POINTER(RtxBuildOptions), C.POINTER(_VP)]),
    "rtx_flat_destroy": (None, [_VP]),
    "rtx_flat_info": (C.c_int32, [_VP, C.POINTER(RtxFlatInfo)]),
    "rtx_scene_upload": (C.c_int32, [_VP, C.POINTER(_VP)]),
    "rtx_scene_upload_f32": (C.c_int32, [_VP, C.POINTER(_VP)]),
    "rtx_scene_is_f32": (C.c_int32, [_VP]),
    "rtx_scene_destroy": (None, [_VP]),
    "rtx_render": (C.c_int32, [_VP, C.POINTER(RtxCamera), C.POINTER(RtxConfig), C.POINTER(RtxFrame)]),
    "rtx_shard_rows": (C.c_int32, [C.POINTER(RtxConfig), C.POINTER(RtxShard)]),
    "rtx_render_device": (C.c_int32, [_VP, C.POINTER(RtxCamera), C.POINTER(RtxConfig), C.POINTER(RtxShard),
                                      _VP, _VP, _VP, C.POINTER(RtxRenderStats)]),
    "rtx_render_count": (C.c_int32, [_VP, C.POINTER(RtxCamera), C.POINTER(RtxConfig), C.POINTER(RtxShard),
                                     C.POINTER(RtxRenderStats)]),
    "rtx_write_ppm": (C.c_int32, [C.c_char_p, C.c_int32, C.c_int32, C.POINTER(C.c_uint8)]),
    "rtx_device_math": (C.c_int32, [C.c_int32, _D3, _D3, C.c_int64, _D3]),
    "rtx_device_stream": (C.c_int32, [C.c_uint64, C.c_uint64, C.c_uint32, C.c_int32, _D3]),
    "rtx_builder_graph": (_VP, [_VP]),
    "rtx_flat_arrays": (_VP, [_VP]),
}


def _load():
    if not os.path.exists(_LIB_PATH):
        raise ImportError(
            "HIP library %s is missing: run `python __graft_entry__.py` (or ray-tracing-series-rust_amd/build.py). "
            "This package has no CPU fallback." % _LIB_PATH)
    lib = C.CDLL(_LIB_PATH)
    for name, (res, args) in ABI.items():
        fn = getattr(lib, name)  # AttributeError here = library/header drift: fail loudly
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()
LIB_PATH = _LIB_PATH


def last_error():
    return (lib.rtx_last_error() or b"").decode("utf-8", "replace")


def _check(status):
    if status != RTX_OK:
        raise RtxError(status, last_error())


def _v3(v):
    a = (C.c_double * 3)(float(v[0]), float(v[1]), float(v[2]))
    return a


def _handle(h):
    if h < 0:
        raise RtxError(RTX_EINVAL, last_error())
    return h


class Camera:
    """Camera::new(lookfrom, lookat, vup, vfov, aspect_ratio, aperture, focus_dist, time1, time2) -- camera.rs:20-57"""

    @staticmethod
    def new(lookfrom, lookat, vup, vfov, aspect_ratio, aperture, focus_dist, time1, time2):
        cam = RtxCamera()
        _check(lib.rtx_camera_new(_v3(lookfrom), _v3(lookat), _v3(vup), vfov, aspect_ratio, aperture, focus_dist,
                                  time1, time2, C.byref(cam)))
        return cam


class Config:
    """Config::new(aspect_ratio, image_width, samples_per_pixel, max_depth, threads) -- world.rs:29-50"""

    @staticmethod
    def new(aspect_ratio, image_width, samples_per_pixel, max_depth, threads, seed=1, background=(0.7, 0.8, 1.0),
            row_chunk_compat=False, sample_buffer_bytes=0):
        cfg = RtxConfig()
        _check(lib.rtx_config_new(aspect_ratio, image_width, samples_per_pixel, max_depth, threads, C.byref(cfg)))
        cfg.seed = seed
        cfg.background[0], cfg.background[1], cfg.background[2] = background
        cfg.row_chunk_compat = 1 if row_chunk_compat else 0
        cfg.sample_buffer_bytes = sample_buffer_bytes
        return cfg


def image_height(cfg):
    return lib.rtx_image_height(C.byref(cfg))


class Builder:
    """Scene construction: one method per reference constructor (argument order kept)."""

    def __init__(self, scene_seed=1):
        p = _VP()
        _check(lib.rtx_builder_create(scene_seed, C.byref(p)))
        self._p = p

    def __del__(self):
        p, self._p = getattr(self, "_p", None), None
        if p:
            lib.rtx_builder_destroy(p)

    @property
    def ptr(self):
        return self._p

    def graph_ptr(self):
        return lib.rtx_builder_graph(self._p)

    def random(self):
        return lib.rtx_builder_random(self._p)

    # textures (texture.rs)
    def solid_color(self, rgb):
        return _handle(lib.rtx_solid_color(self._p, _v3(rgb)))

    def checker(self, even, odd):
        return _handle(lib.rtx_checker(self._p, even, odd))

    def checker_from_colors(self, even_rgb, odd_rgb):  # Checker::from_colors, texture.rs:46-51
        return self.checker(self.solid_color(even_rgb), self.solid_color(odd_rgb))

    def noise(self, scale):
        return _handle(lib.rtx_noise(self._p, scale))

    def image_from_ppm(self, path):
        return _handle(lib.rtx_image_from_ppm(self._p, os.fsencode(path)))

    def image_from_texels(self, texels):
        t = np.ascontiguousarray(texels, dtype=np.float64)
        h, w = t.shape[0], t.shape[1]
        return _handle(lib.rtx_image_from_texels(self._p, w, h, t.ctypes.data_as(_D3)))

    # materials (hit.rs:992-1152)
    def lambertian(self, albedo):
        """Lambertian::new(color) when given an rgb triple, Lambertian::from_pointer(texture) for a handle."""
        tex = albedo if isinstance(albedo, int) else self.solid_color(albedo)
        return _handle(lib.rtx_lambertian(self._p, tex))

    def metal(self, albedo, fuzz):
        return _handle(lib.rtx_metal(self._p, _v3(albedo), fuzz))

    def dielectric(self, ir):
        return _handle(lib.rtx_dielectric(self._p, ir))

    def diffuse_light(self, emit):
        tex = emit if isinstance(emit, int) else self.solid_color(emit)
        return _handle(lib.rtx_diffuse_light(self._p, tex))

    def isotropic(self, albedo):
        tex = albedo if isinstance(albedo, int) else self.solid_color(albedo)
        return _handle(lib.rtx_isotropic(self._p, tex))

    # hittables (hit.rs, bvh.rs, model.rs)
    def sphere(self, center, radius, mat):
        return _handle(lib.rtx_sphere(self._p, _v3(center), radius, mat))

    def moving_sphere(self, center0, center1, time0, time1, radius, mat):
        return _handle(lib.rtx_moving_sphere(self._p, _v3(center0), _v3(center1), time0, time1, radius, mat))

    def gravity_sphere(self, start, time0, radius, mat):  # GravitySphere::new, hit.rs:340-367
        return _handle(lib.rtx_gravity_sphere(self._p, _v3(start), time0, radius, mat))

    def triangle(self, v0, v1, v2, mat):
        return _handle(lib.rtx_triangle(self._p, _v3(v0), _v3(v1), _v3(v2), mat))

    def xy_rect(self, x0, x1, y0, y1, k, mat):
        return _handle(lib.rtx_xy_rect(self._p, x0, x1, y0, y1, k, mat))

    def xz_rect(self, x0, x1, y0, y1, k, mat):
        return _handle(lib.rtx_xz_rect(self._p, x0, x1, y0, y1, k, mat))

    def yz_rect(self, x0, x1, y0, y1, k, mat):
        return _handle(lib.rtx_yz_rect(self._p, x0, x1, y0, y1, k, mat))

    def rect_prism(self, p0, p1, mat):
        return _handle(lib.rtx_rect_prism(self._p, _v3(p0), _v3(p1), mat))

    def hittable_list(self, objects=()):
        lst = _handle(lib.rtx_hittable_list_new(self._p))
        for o in objects:
            self.list_add(lst, o)
        return lst

    def list_add(self, lst, obj):
        _check(lib.rtx_hittable_list_add(self._p, lst, obj))

    def bvh_from_list(self, lst, time0, time1):
        return _handle(lib.rtx_bvh_from_list(self._p, lst, time0, time1))

    def translate(self, offset, obj):
        return _handle(lib.rtx_translate(self._p, _v3(offset), obj))

    def rotate_y(self, angle_degrees, obj):
        return _handle(lib.rtx_rotate_y(self._p, angle_degrees, obj))

    def constant_medium(self, rgb, density, boundary):
        return _handle(lib.rtx_constant_medium(self._p, _v3(rgb), density, boundary))

    def triangle_model(self, path, scale):
        return _handle(lib.rtx_triangle_model(self._p, os.fsencode(path), scale))

    def triangle_mesh(self, vertices, faces, mat):
        v = np.ascontiguousarray(vertices, dtype=np.float64).reshape(-1, 3)
        f = np.ascontiguousarray(faces, dtype=np.int64).reshape(-1, 3)
        return _handle(lib.rtx_triangle_mesh(self._p, v.ctypes.data_as(_D3), v.shape[0],
                                             f.ctypes.data_as(C.POINTER(C.c_int64)), f.shape[0], mat))

    # scene catalogue (world.rs:876-1179)
    def get_world_cam(self, scene_id, camera_aspect=0.0, earth_ppm=None, dragon_ply=None, mesh_triangles=0,
                      book2_boxes_per_side=0, book2_spheres=0):
        opt = RtxSceneOptions(camera_aspect, os.fsencode(earth_ppm) if earth_ppm else None,
                              os.fsencode(dragon_ply) if dragon_ply else None, mesh_triangles,
                              book2_boxes_per_side, book2_spheres)
        world = _H()
        cam = RtxCamera()
        bg = (C.c_double * 3)()
        _check(lib.rtx_get_world_cam(self._p, scene_id, C.byref(opt), C.byref(world), C.byref(cam), bg))
        return world.value, cam, (bg[0], bg[1], bg[2])

    def flatten(self, world, max_leaf=0, sah_bins=0, reference_bvh=False, bvh_seed=0, gpu_builder=False):
        return Flat(self, world, max_leaf, sah_bins, reference_bvh, bvh_seed, gpu_builder)


class Flat:
    """Flattened scene in host memory (rtx_flat)."""

    def __init__(self, builder, world, max_leaf=0, sah_bins=0, reference_bvh=False, bvh_seed=0, gpu_builder=False):
        opt = RtxBuildOptions(max_leaf, sah_bins, 1 if reference_bvh else 0, 1 if gpu_builder else 0, bvh_seed)
        p = _VP()
        _check(lib.rtx_flatten(builder.ptr, world, C.byref(opt), C.byref(p)))
        self._p = p
        self._builder = builder  # keep images/perlin tables alive for the oracle's views

    def __del__(self):
        p, self._p = getattr(self, "_p", None), None
        if p:
            lib.rtx_flat_destroy(p)

    @property
    def ptr(self):
        return self._p

    def arrays_ptr(self):
        return lib.rtx_flat_arrays(self._p)

    def info(self):
        info = RtxFlatInfo()
        _check(lib.rtx_flat_info(self._p, C.byref(info)))
        return {n: getattr(info, n) for n, _ in RtxFlatInfo._fields_}

    def top_level_kinds(self):
        """Entry kinds of the flattened world list, in HittableList order (rtx_flat_top_level_kind)."""
        n = self.info()["n_top_level"]
        return [lib.rtx_flat_top_level_kind(self._p, i) for i in range(n)]

    def upload(self, f32=False):
        """f32=True: the statistical fast mode (rtx_scene_upload_f32): float arithmetic, no bit-exactness claim."""
        return Scene(self, f32=f32)


class Scene:
    """Scene resident on the current HIP device (rtx_scene)."""

    def __init__(self, flat, f32=False):
        p = _VP()
        _check((lib.rtx_scene_upload_f32 if f32 else lib.rtx_scene_upload)(flat.ptr, C.byref(p)))
        self._p = p

    @property
    def is_f32(self):
        return bool(lib.rtx_scene_is_f32(self._p))

    def __del__(self):
        p, self._p = getattr(self, "_p", None), None
        if p:
            lib.rtx_scene_destroy(p)

    @property
    def ptr(self):
        return self._p

    def render(self, cam, cfg, want_accum=True):
        """Whole image on the current device -> Screen (host arrays)."""
        w, h = cfg.image_width, image_height(cfg)
        accum = np.zeros((h, w, 3), dtype=np.float64) if want_accum else None
        rgb8 = np.zeros((h, w, 3), dtype=np.uint8)
        frame = RtxFrame(accum.ctypes.data_as(C.POINTER(C.c_double)) if want_accum else None,
                         rgb8.ctypes.data_as(C.POINTER(C.c_uint8)))
        _check(lib.rtx_render(self._p, C.byref(cam), C.byref(cfg), C.byref(frame)))
        return Screen(w, h, rgb8, accum)

    def trim(self):
        """Release the render workspace (sample buffer, accumulators); the geometry stays resident."""
        _check(lib.rtx_scene_trim(self._p))

    def render_scene_with_time(self, t0, t1, path, row_chunk_compat=True, overrides=None):
        """render_scene_with_time(t0, t1, path, world) of world.rs:1249-1330 on this resident scene: one 500x500 PPM frame."""
        _check(lib.rtx_render_scene_with_time(self._p, t0, t1, path.encode(), 1 if row_chunk_compat else 0,
                                              C.byref(overrides) if overrides is not None else None))

    def render_device(self, cam, cfg, shard=None, d_accum=0, d_rgb8=0, stream=0, want_stats=False):
        """Asynchronous render of one shard into DEVICE buffers (raw pointers, e.g. torch .data_ptr())."""
        sh = RtxShard(*shard, 0) if shard is not None else None
        stats = RtxRenderStats() if want_stats else None
        _check(lib.rtx_render_device(self._p, C.byref(cam), C.byref(cfg), C.byref(sh) if sh else None,
                                     _VP(d_accum or None), _VP(d_rgb8 or None), _VP(stream or None),
                                     C.byref(stats) if stats else None))
        return stats

    def render_count(self, cam, cfg, shard=None):
        sh = RtxShard(*shard, 0) if shard is not None else None
        stats = RtxRenderStats()
        _check(lib.rtx_render_count(self._p, C.byref(cam), C.byref(cfg), C.byref(sh) if sh else None, C.byref(stats)))
        return stats


class MultiScene:
    """The scene resident on several GPUs of this process (rtx_multi): row-interleaved shards, one RCCL gather."""

    def __init__(self, flat, n_shards, device_ids=None, block_rows=1, f32=False):
        ids = (C.c_int32 * n_shards)(*device_ids) if device_ids is not None else None
        p = _VP()
        _check((lib.rtx_multi_create_f32 if f32 else lib.rtx_multi_create)(flat.ptr, n_shards, ids, block_rows, C.byref(p)))
        self._p = p

    def __del__(self):
        p, self._p = getattr(self, "_p", None), None
        if p:
            lib.rtx_multi_destroy(p)

    def render(self, cam, cfg, want_accum=True):
        w, h = cfg.image_width, image_height(cfg)
        accum = np.zeros((h, w, 3), dtype=np.float64) if want_accum else None
        rgb8 = np.zeros((h, w, 3), dtype=np.uint8)
        frame = RtxFrame(accum.ctypes.data_as(C.POINTER(C.c_double)) if want_accum else None,
                         rgb8.ctypes.data_as(C.POINTER(C.c_uint8)))
        stats = RtxMultiStats()
        _check(lib.rtx_multi_render(self._p, C.byref(cam), C.byref(cfg), C.byref(frame), C.byref(stats)))
        screen = Screen(w, h, rgb8, accum)
        screen.stats = stats
        return screen


def render_multi(flat, cam, cfg, n_gpus, want_accum=True):
    """rtx_render_multi: create on devices 0..n_gpus-1, render once, destroy."""
    w, h = cfg.image_width, image_height(cfg)
    accum = np.zeros((h, w, 3), dtype=np.float64) if want_accum else None
    rgb8 = np.zeros((h, w, 3), dtype=np.uint8)
    frame = RtxFrame(accum.ctypes.data_as(C.POINTER(C.c_double)) if want_accum else None,
                     rgb8.ctypes.data_as(C.POINTER(C.c_uint8)))
    _check(lib.rtx_render_multi(flat.ptr, C.byref(cam), C.byref(cfg), n_gpus, C.byref(frame)))
    return Screen(w, h, rgb8, accum)


def trace_kernel_name(kernel):
    return (lib.rtx_trace_kernel_name(kernel) or b"").decode()


def device_math(fn, x, y=None):
    """Evaluate one arithmetic building block on the GPU (see rtx_device_math)."""
    names = {"sin": 0, "cos": 1, "log": 2, "acos": 3, "atan2": 4, "tan": 5, "sqrt": 6, "div": 7, "muladd": 8, "floor": 9, "wide_key": 10}
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.ascontiguousarray(y if y is not None else np.ones_like(x), dtype=np.float64)
    out = np.empty_like(x)
    _check(lib.rtx_device_math(names[fn], x.ctypes.data_as(_D3), y.ctypes.data_as(_D3), x.size, out.ctypes.data_as(_D3)))
    return out


def device_stream(seed, pixel, sample, n):
    out = np.empty(n, dtype=np.float64)
    _check(lib.rtx_device_stream(seed, pixel, sample, n, out.ctypes.data_as(_D3)))
    return out


def shard_rows(cfg, shard):
    sh = RtxShard(*shard, 0)
    return lib.rtx_shard_rows(C.byref(cfg), C.byref(sh))


class Screen:
    """Framebuffer in the reference's layout: row j = 0 is the bottom image row (screen.rs:30-48)."""

    def __init__(self, width, height, rgb8, accum=None):
        self.width, self.height, self.rgb8, self.accum = width, height, rgb8, accum

    def write_to_ppm_file(self, path):
        buf = np.ascontiguousarray(self.rgb8)
        _check(lib.rtx_write_ppm(os.fsencode(path), self.width, self.height, buf.ctypes.data_as(C.POINTER(C.c_uint8))))

    def write_to_ppm(self):
        sys.stdout.flush()
        buf = np.ascontiguousarray(self.rgb8)
        _check(lib.rtx_write_ppm(None, self.width, self.height, buf.ctypes.data_as(C.POINTER(C.c_uint8))))


def render_scene(builder, world, cam, background, config, max_leaf=0):
    """render_scene(world, cam, background, config) -- world.rs:1181-1247, on the current GPU.

    Returns the Screen instead of printing it; call .write_to_ppm() for the reference's stdout output.
    """
    cfg = RtxConfig.from_buffer_copy(config)
    cfg.background[0], cfg.background[1], cfg.background[2] = background
    scene = builder.flatten(world, max_leaf=max_leaf).upload()
    return scene.render(cam, cfg)
