"""ORACLE -- TEST INFRASTRUCTURE.  ctypes face of oracle/_build/liboracle.so.

Importers allowed: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.  Never the
product package.  Build with `make -C oracle` (or __graft_entry__.build()).
"""
import ctypes as C
import os

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_DIR, "_build", "liboracle.so")


class OracleCamera(C.Structure):
    _fields_ = [(n, C.c_double * 3) for n in ("origin", "lower_left_corner", "horizontal", "vertical", "u", "v", "w")] + \
               [("lens_radius", C.c_double), ("time1", C.c_double), ("time2", C.c_double)]


class OracleConfig(C.Structure):
    _fields_ = [("image_width", C.c_int32), ("image_height", C.c_int32), ("samples_per_pixel", C.c_int32),
                ("max_depth", C.c_int32), ("threads", C.c_int32), ("row_chunk_compat", C.c_int32),
                ("seed", C.c_uint64), ("bvh_seed", C.c_uint64), ("background", C.c_double * 3)]


class OracleCounters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("box_tests", "sphere_tests", "moving_sphere_tests", "rect_tests",
                                          "triangle_tests", "scatters", "texels", "perlin_calls", "rays", "samples")]


_D = C.POINTER(C.c_double)
_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("oracle library %s missing: run `make -C oracle`" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    lib.oracle_o1_render.restype = C.c_int
    lib.oracle_o1_render.argtypes = [C.c_void_p, C.c_int32, C.POINTER(OracleCamera), C.POINTER(OracleConfig), _D, C.POINTER(C.c_uint8)]
    lib.oracle_o2_render.restype = C.c_int
    lib.oracle_o2_render.argtypes = [C.c_void_p, C.POINTER(OracleCamera), C.POINTER(OracleConfig), C.c_int32, C.c_int32,
                                     C.c_int32, _D, C.POINTER(C.c_uint8), C.POINTER(OracleCounters)]
    lib.oracle_o2_sample.restype = C.c_int
    lib.oracle_o2_sample.argtypes = [C.c_void_p, C.POINTER(OracleCamera), C.POINTER(OracleConfig), C.c_int32, C.c_int32, C.c_int32, _D]
    lib.oracle_o1_vec3_ops.restype = None
    lib.oracle_o1_vec3_ops.argtypes = [_D, _D, C.c_double, _D]
    lib.oracle_o1_tone_map.restype = None
    lib.oracle_o1_tone_map.argtypes = [_D, C.c_uint32, C.POINTER(C.c_int32)]
    lib.oracle_o1_sphere_uv.restype = None
    lib.oracle_o1_sphere_uv.argtypes = [_D, _D]
    lib.oracle_o1_reflectance.restype = C.c_double
    lib.oracle_o1_reflectance.argtypes = [C.c_double, C.c_double]
    lib.oracle_o1_refract.restype = None
    lib.oracle_o1_refract.argtypes = [_D, _D, C.c_double, _D]
    lib.oracle_o1_reflect.restype = None
    lib.oracle_o1_reflect.argtypes = [_D, _D, _D]
    lib.oracle_o1_aabb_hit.restype = C.c_int
    lib.oracle_o1_aabb_hit.argtypes = [_D, _D, _D, _D, C.c_double, C.c_double]
    lib.oracle_o1_hit.restype = C.c_int
    lib.oracle_o1_hit.argtypes = [C.c_void_p, C.c_int32, _D, _D, C.c_double, C.c_double, C.c_double, C.c_uint64, _D]
    for side in ("o1", "core"):
        getattr(lib, "oracle_%s_vec3_ops" % side).restype = None
        getattr(lib, "oracle_%s_vec3_ops" % side).argtypes = [_D, _D, C.c_double, _D]
        getattr(lib, "oracle_%s_tone_map" % side).restype = None
        getattr(lib, "oracle_%s_tone_map" % side).argtypes = [_D, C.c_uint32, C.POINTER(C.c_int32)]
        getattr(lib, "oracle_%s_sphere_uv" % side).restype = None
        getattr(lib, "oracle_%s_sphere_uv" % side).argtypes = [_D, _D]
        getattr(lib, "oracle_%s_reflectance" % side).restype = C.c_double
        getattr(lib, "oracle_%s_reflectance" % side).argtypes = [C.c_double, C.c_double]
        getattr(lib, "oracle_%s_refract" % side).restype = None
        getattr(lib, "oracle_%s_refract" % side).argtypes = [_D, _D, C.c_double, _D]
        getattr(lib, "oracle_%s_reflect" % side).restype = None
        getattr(lib, "oracle_%s_reflect" % side).argtypes = [_D, _D, _D]
        getattr(lib, "oracle_%s_aabb_hit" % side).restype = C.c_int
        getattr(lib, "oracle_%s_aabb_hit" % side).argtypes = [_D, _D, _D, _D, C.c_double, C.c_double]
    lib.oracle_core_world_hit.restype = C.c_int
    lib.oracle_core_world_hit.argtypes = [C.c_void_p, _D, _D, C.c_double, C.c_double, C.c_double, C.c_uint64, _D]
    lib.oracle_audit_flat.restype = C.c_int
    lib.oracle_audit_flat.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
    lib.oracle_audit_motion.restype = C.c_int
    lib.oracle_audit_motion.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int64)]
    lib.oracle_motion_leaf_area_ratio.restype = C.c_double
    lib.oracle_motion_leaf_area_ratio.argtypes = [C.c_void_p]
    lib.oracle_philox4x32_10.restype = None
    lib.oracle_philox4x32_10.argtypes = [C.POINTER(C.c_uint32), C.c_uint32, C.c_uint32]
    lib.oracle_splitmix64_next.restype = C.c_uint64
    lib.oracle_splitmix64_next.argtypes = [C.POINTER(C.c_uint64)]
    lib.oracle_sample_stream.restype = None
    lib.oracle_sample_stream.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_int32, _D]
    lib.oracle_rt_math.restype = None
    lib.oracle_rt_math.argtypes = [C.c_int32, _D, _D, C.c_int64, _D]
    _lib = lib
    return lib


def _d3(v):
    return (C.c_double * 3)(float(v[0]), float(v[1]), float(v[2]))


def camera_from(rtx_cam):
    """RtxCamera (product struct) -> OracleCamera (same layout)."""
    return OracleCamera.from_buffer_copy(bytes(rtx_cam))


def config_from(rtx_cfg, image_height, threads=None, bvh_seed=12345):
    c = OracleConfig()
    c.image_width = rtx_cfg.image_width
    c.image_height = image_height
    c.samples_per_pixel = rtx_cfg.samples_per_pixel
    c.max_depth = rtx_cfg.max_depth
    c.threads = threads if threads is not None else rtx_cfg.threads
    c.row_chunk_compat = rtx_cfg.row_chunk_compat
    c.seed = rtx_cfg.seed
    c.bvh_seed = bvh_seed
    c.background[0], c.background[1], c.background[2] = rtx_cfg.background[0], rtx_cfg.background[1], rtx_cfg.background[2]
    return c


def o1_render(graph_ptr, world, rtx_cam, rtx_cfg, image_height, threads=None, bvh_seed=12345):
    """Literal restatement.  Returns (accum[h,w,3] f64, rgb8[h,w,3] u8), row 0 = bottom row."""
    lib = load()
    cam = camera_from(rtx_cam)
    cfg = config_from(rtx_cfg, image_height, threads, bvh_seed)
    h, w = image_height, rtx_cfg.image_width
    accum = np.zeros((h, w, 3), dtype=np.float64)
    rgb8 = np.zeros((h, w, 3), dtype=np.uint8)
    rc = lib.oracle_o1_render(graph_ptr, world, C.byref(cam), C.byref(cfg), accum.ctypes.data_as(_D),
                              rgb8.ctypes.data_as(C.POINTER(C.c_uint8)))
    if rc != 0:
        raise RuntimeError("oracle_o1_render failed (%d)" % rc)
    return accum, rgb8


def o1_render_window(graph_ptr, world, rtx_cam, rtx_cfg, image_height, top_box, threads=None, bvh_seed=12345):
    """The literal restatement over a rectangle of the frame.  top_box = (r0, r1, c0, c1) with rows counted from the TOP of the
    image (as the reference-image pins are); returns accum[r1 - r0, c1 - c0, 3], top row first."""
    lib = load()
    cam = camera_from(rtx_cam)
    cfg = config_from(rtx_cfg, image_height, threads, bvh_seed)
    r0, r1, c0, c1 = top_box
    j0, j1 = image_height - r1, image_height - r0
    accum = np.zeros((j1 - j0, c1 - c0, 3), dtype=np.float64)
    lib.oracle_o1_render_window.restype = C.c_int
    rc = lib.oracle_o1_render_window(C.c_void_p(graph_ptr), C.c_int32(world), C.byref(cam), C.byref(cfg), C.c_int32(j0), C.c_int32(j1),
                                     C.c_int32(c0), C.c_int32(c1), accum.ctypes.data_as(_D))
    if rc != 0:
        raise RuntimeError("oracle_o1_render_window failed (%d)" % rc)
    return accum[::-1]


def o2_render(flat_arrays_ptr, rtx_cam, rtx_cfg, image_height, shard=(0, 1, 1), threads=None, counters=False):
    """Flat-array loop through the product's core headers on the CPU."""
    lib = load()
    cam = camera_from(rtx_cam)
    cfg = config_from(rtx_cfg, image_height, threads)
    w = rtx_cfg.image_width
    rows = [j for j in range(image_height) if (j // shard[2]) % shard[1] == shard[0]]
    accum = np.zeros((len(rows), w, 3), dtype=np.float64)
    rgb8 = np.zeros((len(rows), w, 3), dtype=np.uint8)
    cnt = OracleCounters() if counters else None
    rc = lib.oracle_o2_render(flat_arrays_ptr, C.byref(cam), C.byref(cfg), shard[0], shard[1], shard[2],
                              accum.ctypes.data_as(_D), rgb8.ctypes.data_as(C.POINTER(C.c_uint8)),
                              C.byref(cnt) if cnt is not None else None)
    if rc != 0:
        raise RuntimeError("oracle_o2_render failed (%d)" % rc)
    if counters:
        return accum, rgb8, {n: getattr(cnt, n) for n, _ in OracleCounters._fields_}
    return accum, rgb8


def o2_sample(flat_arrays_ptr, rtx_cam, rtx_cfg, image_height, i, j, s):
    lib = load()
    cam = camera_from(rtx_cam)
    cfg = config_from(rtx_cfg, image_height)
    out = (C.c_double * 3)()
    lib.oracle_o2_sample(flat_arrays_ptr, C.byref(cam), C.byref(cfg), i, j, s, out)
    return np.array(out[:])


def rt_math(fn, x, y=None):
    lib = load()
    names = {"sin": 0, "cos": 1, "log": 2, "acos": 3, "atan2": 4, "tan": 5, "sqrt": 6, "sin_sign": 10}
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.ascontiguousarray(y if y is not None else np.zeros_like(x), dtype=np.float64)
    out = np.empty_like(x)
    lib.oracle_rt_math(names[fn], x.ctypes.data_as(_D), y.ctypes.data_as(_D), x.size, out.ctypes.data_as(_D))
    return out


# ---- thin probe wrappers used by the known-answer tests (side = "o1" literal restatement, "core" product core)
def vec3_ops(side, a, b, t):
    out = (C.c_double * 24)()
    getattr(load(), "oracle_%s_vec3_ops" % side)(_d3(a), _d3(b), float(t), out)
    o = list(out)
    names = ["add", "sub", "mul", "mul_t", "div_t", "neg", "cross"]
    res = {n: tuple(o[3 * i:3 * i + 3]) for i, n in enumerate(names)}
    res["dot"], res["length_squared"], res["length"] = o[21], o[22], o[23]
    return res


def tone_map(side, s, spp):
    out = (C.c_int32 * 3)()
    getattr(load(), "oracle_%s_tone_map" % side)(_d3(s), spp, out)
    return tuple(out)


def sphere_uv(side, p):
    out = (C.c_double * 2)()
    getattr(load(), "oracle_%s_sphere_uv" % side)(_d3(p), out)
    return out[0], out[1]


def reflectance(side, cosine, ref_idx):
    return getattr(load(), "oracle_%s_reflectance" % side)(cosine, ref_idx)


def refract(side, uv, n, ratio):
    out = (C.c_double * 3)()
    getattr(load(), "oracle_%s_refract" % side)(_d3(uv), _d3(n), ratio, out)
    return tuple(out)


def reflect(side, v, n):
    out = (C.c_double * 3)()
    getattr(load(), "oracle_%s_reflect" % side)(_d3(v), _d3(n), out)
    return tuple(out)


def aabb_hit(side, mn, mx, o, d, t_min, t_max):
    return bool(getattr(load(), "oracle_%s_aabb_hit" % side)(_d3(mn), _d3(mx), _d3(o), _d3(d), t_min, t_max))


def _rec(rc, out):
    if rc <= 0:
        return None
    o = list(out)
    return {"t": o[0], "p": tuple(o[1:4]), "normal": tuple(o[4:7]), "u": o[7], "v": o[8], "front_face": bool(o[9])}


def o1_hit(graph_ptr, handle, o, d, time=0.0, t_min=0.001, t_max=float("inf"), rng_seed=1):
    out = (C.c_double * 10)()
    return _rec(load().oracle_o1_hit(graph_ptr, handle, _d3(o), _d3(d), time, t_min, t_max, rng_seed, out), out)


def core_world_hit(flat_arrays_ptr, o, d, time=0.0, t_min=0.001, t_max=float("inf"), rng_seed=1):
    out = (C.c_double * 10)()
    return _rec(load().oracle_core_world_hit(flat_arrays_ptr, _d3(o), _d3(d), time, t_min, t_max, rng_seed, out), out)


def audit_motion(flat_arrays_ptr, n_times=16):
    """(code, triples checked): the time-aware boxes contain every sphere below them at n_times instants (0 = consistent)."""
    n = C.c_int64(0)
    rc = load().oracle_audit_motion(flat_arrays_ptr, n_times, C.byref(n))
    return rc, n.value


def lds_walk_render(flat_arrays_ptr, rtx_cam, rtx_cfg, image_height, use_motion, row_stride=1, threads=None):
    """The LDS kernel's walk on the CPU (f32 culling tree, static or time-aware boxes): (accum of rows 0, stride, .., counts)."""
    lib = load()
    cam = camera_from(rtx_cam)
    cfg = config_from(rtx_cfg, image_height, threads)
    rows = list(range(0, image_height, row_stride))
    accum = np.zeros((len(rows), rtx_cfg.image_width, 3), dtype=np.float64)
    counts = (C.c_uint64 * 3)()
    lib.oracle_lds_walk_render.restype = C.c_int
    rc = lib.oracle_lds_walk_render(C.c_void_p(flat_arrays_ptr), C.byref(cam), C.byref(cfg), C.c_int32(1 if use_motion else 0),
                                    C.c_int32(row_stride), accum.ctypes.data_as(_D), counts)
    if rc != 0:
        raise RuntimeError("oracle_lds_walk_render failed (%d)" % rc)
    return accum, {"rays": counts[0], "node_visits": counts[1], "prim_tests": counts[2]}


def motion_leaf_area_ratio(flat_arrays_ptr):
    return load().oracle_motion_leaf_area_ratio(flat_arrays_ptr)


def audit_flat(flat_arrays_ptr):
    depth = C.c_int32(0)
    rc = load().oracle_audit_flat(flat_arrays_ptr, C.byref(depth))
    return rc, depth.value
