/* rtx_abi.h -- C ABI of the MI355X-native path-tracing hot path.
 *
 * The reference (patrickzbhe/ray-tracing-series-rust) has no FFI, plugin or operator
 * interface.  Its one seam around the hot path is
 *
 *     pub fn render_scene(world: Arc<Box<dyn Hittable + Sync>>, cam: Arc<Camera>,
 *                         background: Vec3, config: Config)        src/world.rs:1181-1186
 *
 * called from src/main.rs:13 after get_world_cam() (src/world.rs:876).  `dyn Hittable`
 * exposes only hit()/bounding_box() and all struct fields are private, so a Rust host can
 * hand its scene across a boundary only by walking its own constructors.  This header is
 * that boundary: one builder entry point per reference constructor (same argument order),
 * a flatten + upload step, and the render call that replaces render_scene's per-pixel x
 * per-sample loop (src/world.rs:1207-1226) with HIP kernels on gfx950.
 *
 * Conventions: C99, no exceptions or aborts cross this boundary; every call that can fail
 * returns an rtx_status (or a negative handle) and leaves a message for rtx_last_error().
 * Plain pointers and sizes only.  Structs are POD, little-endian, 8-byte aligned.
 * Handles (rtx_handle) are indices owned by one builder.  A builder is single-threaded.  An rtx_flat is immutable
 * after creation and may be shared between threads.  An rtx_scene's GEOMETRY is immutable, but the handle also owns the
 * render workspace (sample buffer, accumulators, work counter, timers): at most ONE render call may be in flight per
 * rtx_scene at a time -- calls on one handle must come from one thread at a time and, for rtx_render_device, on one
 * stream; concurrent renders of the same scene need one rtx_scene each (rtx_scene_upload is cheap next to a render).
 * The workspace is kept until rtx_scene_destroy or rtx_scene_trim (default budget of the sample buffer: 24 GiB).
 */
#ifndef RTX_ABI_H
#define RTX_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTX_ABI_VERSION 1

typedef int32_t rtx_status;
#define RTX_OK 0
#define RTX_EINVAL 1       /* bad argument; also the reference's assert!/panic! conditions */
#define RTX_ENOMEM 2
#define RTX_EHIP 3         /* a HIP runtime call failed (includes "no GPU") */
#define RTX_EUNSUPPORTED 4 /* scene shape outside what the kernels implement */
#define RTX_EIO 5          /* file could not be read/written (reference: expect()/unwrap() panics) */
#define RTX_ENCCL 6        /* RCCL could not be loaded, or a collective failed */

typedef int32_t rtx_handle; /* >= 0 valid, < 0 error */

typedef struct rtx_builder rtx_builder; /* scene graph under construction (host only) */
typedef struct rtx_flat rtx_flat;       /* flattened scene arrays + BVHs (host memory) */
typedef struct rtx_scene rtx_scene;     /* flattened scene resident in one GPU's HBM */

/* ---- library ------------------------------------------------------------------------ */
int32_t rtx_abi_version(void);
/* Thread-local; valid until the next failing call on the same thread. */
const char* rtx_last_error(void);

/* ---- scene construction: one entry point per reference constructor -------------------- */
/* scene_seed seeds the construction-time random stream (Perlin tables, catalogue scenes);
 * the reference uses rand::thread_rng() there. */
rtx_status rtx_builder_create(uint64_t scene_seed, rtx_builder** out);
void rtx_builder_destroy(rtx_builder* b); /* NULL-safe */
/* One uniform f64 in [0,1) from the construction stream (for host-side scene generators). */
double rtx_builder_random(rtx_builder* b);

/* Textures -- src/texture.rs */
rtx_handle rtx_solid_color(rtx_builder* b, const double rgb[3]);                 /* SolidColor::new            texture.rs:16-20  */
rtx_handle rtx_checker(rtx_builder* b, rtx_handle even, rtx_handle odd);        /* Checker::new               texture.rs:39-44  */
rtx_handle rtx_noise(rtx_builder* b, double scale);                             /* Noise::new (+Perlin::new)  texture.rs:72-77  */
rtx_handle rtx_image_from_ppm(rtx_builder* b, const char* path);                /* Image::from_ppm            texture.rs:95-99  */
rtx_handle rtx_image_from_texels(rtx_builder* b, int32_t width, int32_t height,
                                 const double* rgb_0_255);                      /* Screen contents, row 0 = first PPM row */
/* Materials -- src/hit.rs:992-1152 */
rtx_handle rtx_lambertian(rtx_builder* b, rtx_handle texture);                  /* Lambertian::from_pointer   hit.rs:1031-1035 */
rtx_handle rtx_metal(rtx_builder* b, const double albedo[3], double fuzz);      /* Metal::new                 hit.rs:1060-1065 */
rtx_handle rtx_dielectric(rtx_builder* b, double ir);                           /* Dielectric::new            hit.rs:1091-1093 */
rtx_handle rtx_diffuse_light(rtx_builder* b, rtx_handle texture);               /* DiffuseLight::from_pointer hit.rs:1140-1142 */
rtx_handle rtx_isotropic(rtx_builder* b, rtx_handle texture);                   /* Isotropic::from_color      hit.rs:997-1001  */
/* Hittables -- src/hit.rs, src/bvh.rs, src/model.rs */
rtx_handle rtx_sphere(rtx_builder* b, const double center[3], double radius, rtx_handle mat);          /* hit.rs:187-193 */
rtx_handle rtx_moving_sphere(rtx_builder* b, const double center0[3], const double center1[3],
                             double time0, double time1, double radius, rtx_handle mat);               /* hit.rs:257-273 */
/* GravitySphere::new(start, time0, radius, mat): the bouncing ball of the video scene; simulates and stores its ~100 002
 * heights at construction, as the reference does. */
rtx_handle rtx_gravity_sphere(rtx_builder* b, const double start[3], double time0, double radius, rtx_handle mat); /* hit.rs:340-367 */
rtx_handle rtx_triangle(rtx_builder* b, const double v0[3], const double v1[3], const double v2[3],
                        rtx_handle mat);                                                                /* hit.rs:96-107  */
rtx_handle rtx_xy_rect(rtx_builder* b, double x0, double x1, double y0, double y1, double k, rtx_handle mat); /* hit.rs:456-472 */
rtx_handle rtx_xz_rect(rtx_builder* b, double x0, double x1, double y0, double y1, double k, rtx_handle mat); /* hit.rs:521-537 */
rtx_handle rtx_yz_rect(rtx_builder* b, double x0, double x1, double y0, double y1, double k, rtx_handle mat); /* hit.rs:586-602 */
rtx_handle rtx_rect_prism(rtx_builder* b, const double p0[3], const double p1[3], rtx_handle mat);     /* hit.rs:720-775 */
rtx_handle rtx_hittable_list_new(rtx_builder* b);                                                       /* hit.rs:646-648 */
rtx_status rtx_hittable_list_add(rtx_builder* b, rtx_handle list, rtx_handle object);                   /* hit.rs:650-652 */
rtx_handle rtx_bvh_from_list(rtx_builder* b, rtx_handle list, double time0, double time1);              /* bvh.rs:85-93   */
rtx_handle rtx_translate(rtx_builder* b, const double offset[3], rtx_handle object);                    /* hit.rs:793-798 */
rtx_handle rtx_rotate_y(rtx_builder* b, double angle_degrees, rtx_handle object);                       /* hit.rs:843-888 */
rtx_handle rtx_constant_medium(rtx_builder* b, const double rgb[3], double density, rtx_handle boundary); /* hit.rs:945-951 */
/* TriangleModel::load_from_file(path, scale).to_hittable() -> a HittableList of triangles (model.rs:13-76). */
rtx_handle rtx_triangle_model(rtx_builder* b, const char* path, double scale);
/* Same from memory: vertices = 3 doubles each, faces = 3 vertex indices each. */
rtx_handle rtx_triangle_mesh(rtx_builder* b, const double* vertices, int64_t n_vertices,
                             const int64_t* faces, int64_t n_faces, rtx_handle mat);

/* ---- camera and config --------------------------------------------------------------- */
/* The ten derived fields of src/camera.rs:6-17, as Camera::new computes them. */
typedef struct RtxCamera {
  double origin[3], lower_left_corner[3], horizontal[3], vertical[3], u[3], v[3], w[3];
  double lens_radius, time1, time2;
} RtxCamera;
/* Camera::new(lookfrom, lookat, vup, vfov, aspect_ratio, aperture, focus_dist, time1, time2)  camera.rs:20-57.
 * RTX_EINVAL if time1 >= time2 (gen_range(time1..time2) panics on an empty range, camera.rs:69). */
rtx_status rtx_camera_new(const double lookfrom[3], const double lookat[3], const double vup[3],
                          double vfov_degrees, double aspect_ratio, double aperture,
                          double focus_dist, double time1, double time2, RtxCamera* out);

/* Config::new(aspect_ratio, image_width, samples_per_pixel, max_depth, threads)  world.rs:20-50,
 * followed by the fields this build adds. */
typedef struct RtxConfig {
  double aspect_ratio;
  int32_t image_width;
  int32_t samples_per_pixel;
  int32_t max_depth;
  int32_t threads;          /* kept for API parity; only used by row_chunk_compat */
  uint64_t seed;            /* render seed of the counter-based per-(pixel,sample) streams */
  double background[3];     /* render_scene's `background` argument */
  int32_t row_chunk_compat; /* 1: leave rows >= threads*floor(h/threads) black (world.rs:1198-1202 quirk) */
  int32_t reserved;
  uint64_t sample_buffer_bytes; /* cap on the per-pass sample-radiance buffer; 0 = default */
} RtxConfig;
/* Fills the five reference fields, asserting what Config::new asserts (-> RTX_EINVAL), and
 * defaults: seed 1, background (0.7,0.8,1.0), row_chunk_compat 0. */
rtx_status rtx_config_new(double aspect_ratio, int32_t image_width, int32_t samples_per_pixel,
                          int32_t max_depth, int32_t threads, RtxConfig* out);
/* image_height = (image_width as f64 / aspect_ratio) as i32   world.rs:1192 */
int32_t rtx_image_height(const RtxConfig* cfg);

/* ---- scene catalogue: get_world_cam(config_num)  world.rs:876-1179 -------------------- */
typedef struct RtxSceneOptions {
  double camera_aspect;     /* <= 0: the reference's hard-coded aspect for that scene */
  const char* earth_ppm;    /* NULL or unreadable: procedural stand-in for "earthshit.ppm" */
  const char* dragon_ply;   /* NULL or unreadable: procedural stand-in mesh */
  int64_t mesh_triangles;   /* procedural mesh size; 0 = 871200 */
  int32_t book2_boxes_per_side; /* 0 = 20 */
  int32_t book2_spheres;        /* 0 = 1000 */
} RtxSceneOptions;
/* scene_id 0..12 and "anything else" as in the reference; 100 = canonical Book-1 final scene,
 * 101 = empty world.  options may be NULL. */
rtx_status rtx_get_world_cam(rtx_builder* b, int32_t scene_id, const RtxSceneOptions* options,
                             rtx_handle* world_out, RtxCamera* cam_out, double background_out[3]);

/* ---- flatten + upload ------------------------------------------------------------------ */
typedef struct RtxBuildOptions {
  int32_t max_leaf;      /* primitives per BVH leaf, 1..8; 0 = default (1; 2 for BVHs holding triangles) */
  int32_t sah_bins;      /* 0 = default */
  int32_t reference_bvh; /* 1: build BVHs with the reference's rule (bvh.rs:14-83) instead of SAH: same image,
                            different traversal statistics (A/B switch) */
  int32_t gpu_builder;   /* 1: BVHs of >= 1024 primitives are built on the CURRENT GPU (Morton clusters + radix tree + refit):
                            same image, milliseconds instead of a second for 871 200 triangles (replaces bvh.rs:14-83 at mesh
                            scale); rtx_flatten then needs a GPU (RTX_EUNSUPPORTED if the build cannot run) */
  uint64_t bvh_seed;     /* stream for the reference rule's random split axis */
} RtxBuildOptions;
typedef struct RtxFlatInfo {
  int64_t n_spheres, n_moving_spheres, n_rects, n_triangles;
  int64_t n_nodes, n_refs, n_entries, n_top_level, n_materials, n_textures, n_perlins, n_images, n_texels;
  int64_t total_bytes;
  int32_t max_stack, n_bvh;
  double sah_cost;
  double bvh_build_ms;  /* wall time of all BVH builds of this flatten (host SAH, reference rule or GPU builder) */
  double bvh_device_ms; /* GPU builder: device time by HIP events, box upload and node download included */
  int64_t n_gravity_spheres;
} RtxFlatInfo;
/* Host only (no GPU needed).  options may be NULL. */
rtx_status rtx_flatten(const rtx_builder* b, rtx_handle world, const RtxBuildOptions* options,
                       rtx_flat** out);
void rtx_flat_destroy(rtx_flat* f); /* NULL-safe */
rtx_status rtx_flat_info(const rtx_flat* f, RtxFlatInfo* out);
/* Kind of the index-th slot of the flattened world list (HittableList order, hit.rs:650-652): 0 primitive,
 * 1 ordered group (HittableList / RectPrism), 2 BVH, 3 Translate/RotateY chain, 4 ConstantMedium; -1 = index out
 * of range.  Inspection only (lets a host check that its scene flattened to the list it built). */
int32_t rtx_flat_top_level_kind(const rtx_flat* f, int32_t index);
/* Copies every array to the CURRENT HIP device. */
rtx_status rtx_scene_upload(const rtx_flat* f, rtx_scene** out);
/* The statistical fast mode.  The same scene, narrowed field by field to single precision, rendered by
 * the same kernels compiled with float arithmetic (csrc/hip/render_f32.hip): the image converges to the same picture
 * but is NOT bit-comparable with the reference's -- the bit-exactness this header promises elsewhere is about scenes
 * uploaded with rtx_scene_upload.  Every render entry point takes either kind of scene (rtx_multi_create_f32 for
 * several GPUs); rtx_render_count is f64 only.  The RNG stream per (pixel, sample), the sample order of the sums and the f64 accumulators
 * handed back are the same in both modes. */
rtx_status rtx_scene_upload_f32(const rtx_flat* f, rtx_scene** out);
int32_t rtx_scene_is_f32(const rtx_scene* s);
void rtx_scene_destroy(rtx_scene* s); /* NULL-safe */
/* Releases the render workspace of an idle scene (it is re-allocated by the next render); the geometry stays resident. */
rtx_status rtx_scene_trim(rtx_scene* s);

/* ---- render: replaces render_scene's sample loop ------------------------------------------- */
/* Pixel order everywhere: row-major, row j = 0 is the BOTTOM image row (Screen::update(j, i),
 * world.rs:1235; PPM output walks j downwards, screen.rs:43). */
typedef struct RtxFrame {
  double* accum_rgb; /* optional: h*w*3 per-pixel radiance sums over samples (before tone map) */
  uint8_t* rgb8;     /* optional: h*w*3 tone-mapped channels, get_normalized_color (vec3.rs:89-107) */
} RtxFrame;
/* Counters of one render (filled when requested; counting runs a separate instrumented kernel). */
typedef struct RtxRenderStats {
  uint64_t samples, rays, box_tests, sphere_tests, moving_sphere_tests, rect_tests, triangle_tests;
  uint64_t scatters, texels, perlin_calls;
  double trace_ms, reduce_ms, tonemap_ms; /* device time of the kernels of this render (HIP events) */
  int32_t trace_launches, passes;
  uint64_t sample_buffer_bytes;
  int32_t trace_kernel; /* which trace kernel the launcher chose: RTX_KERNEL_* (rtx_trace_kernel_name) */
  int32_t reserved;
} RtxRenderStats;
/* Trace kernels (all produce identical results; the launcher picks by world shape and LDS budget). */
enum {
  RTX_KERNEL_SIMPLE = 0,     /* grid-stride, one whole path per thread (also the counting kernel) */
  RTX_KERNEL_PERSISTENT = 1, /* persistent waves + path regeneration, wave-synchronous list scan; any world (A/B partner of WORLD) */
  RTX_KERNEL_STREAM = 2,     /* A/B only */
  RTX_KERNEL_VOTE = 3,       /* worlds that are one BVH: node/leaf voting walk, f32 culling, carry-over */
  RTX_KERNEL_LDS = 4,        /* RTX_KERNEL_VOTE with the geometry resident in LDS (sphere worlds that fit) */
  RTX_KERNEL_WQ = 5,         /* experimental (not in the default build): workgroup-level path queues in LDS */
  RTX_KERNEL_WORLD = 6,      /* any world: per-lane scan of the world list, walks of every BVH entry carried over */
  RTX_KERNEL_WAVEFRONT = 7   /* split-kernel integrator: path state in HBM, k_wf_generate / k_wf_trace / k_wf_shade per bounce */
};
const char* rtx_trace_kernel_name(int32_t kernel);
/* Blocking; host output buffers.  Renders the whole image on the current device. */
rtx_status rtx_render(const rtx_scene* s, const RtxCamera* cam, const RtxConfig* cfg, RtxFrame* out);

/* A shard = the image rows { j : (j / block_rows) % shard_count == shard_index }, compacted in
 * ascending j.  shard_count = 1 is the whole image. */
typedef struct RtxShard {
  int32_t shard_index, shard_count, block_rows, reserved;
} RtxShard;
int32_t rtx_shard_rows(const RtxConfig* cfg, const RtxShard* shard); /* number of rows in the shard */
/* Asynchronous on `hip_stream` (a hipStream_t, may be NULL = default stream); outputs are DEVICE
 * pointers sized for the shard (rows*w*3).  d_accum_rgb / d_rgb8 may each be NULL.  Scratch comes
 * from a per-scene workspace that grows on demand (the only call here that may allocate).
 * stats may be NULL; if non-NULL the call synchronises the stream to read timers. */
rtx_status rtx_render_device(const rtx_scene* s, const RtxCamera* cam, const RtxConfig* cfg,
                             const RtxShard* shard, double* d_accum_rgb, uint8_t* d_rgb8,
                             void* hip_stream, RtxRenderStats* stats);
/* Instrumented render of the same shard: fills the work counters of RtxRenderStats (blocking). */
rtx_status rtx_render_count(const rtx_scene* s, const RtxCamera* cam, const RtxConfig* cfg,
                            const RtxShard* shard, RtxRenderStats* stats);

/* ---- one process, several GPUs: replaces the band threads + collect loop of render_scene (world.rs:1198-1244) ---- */
/* The frame is cut into n_shards row-interleaved shards (RtxShard with shard_count = n_shards); shard r is rendered
 * by device device_ids[r] (NULL: device r) into that device's HBM, then ONE RCCL gather (ncclGather over xGMI, a
 * communicator from ncclCommInitAll) brings the shards to the first device, where they are put in row order and
 * copied to the host buffers of `out`.  The scene is uploaded to every device once, at creation, and stays resident
 * across renders.  device_ids must be all distinct -- or all equal, which renders the shards one after another on
 * that one GPU without the collective (a rehearsal of the sharding on a single-GPU box).  The frame is byte-identical
 * to rtx_render's for every shard count.  n_shards = 1 with one device goes through RCCL as well. */
typedef struct rtx_multi rtx_multi;
typedef struct RtxMultiStats {
  double render_ms_max;   /* slowest device, first launch to end of tone map (HIP events) */
  double total_ms;        /* host wall time of the call: launches + gather + reorder + copy to the host */
  uint64_t gathered_bytes;
  int32_t n_shards, n_devices, used_rccl;
  int32_t rccl_ranks;     /* ncclCommCount of the communicator the gather ran on (0 without one) */
  double gather_ms;       /* on the first device's stream: end of its own shard -> every shard gathered and put in row order
                             (includes waiting for the slowest peer) */
  double render_ms[16];   /* per device (first 16), as render_ms_max */
} RtxMultiStats;
rtx_status rtx_multi_create(const rtx_flat* f, int32_t n_shards, const int32_t* device_ids, int32_t block_rows,
                            rtx_multi** out);
/* The same with every shard's scene uploaded by rtx_scene_upload_f32 (the statistical fast mode). */
rtx_status rtx_multi_create_f32(const rtx_flat* f, int32_t n_shards, const int32_t* device_ids, int32_t block_rows,
                                rtx_multi** out);
void rtx_multi_destroy(rtx_multi* m); /* NULL-safe */
/* Blocking.  out->rgb8 and / or out->accum_rgb: host buffers of h*w*3 elements (row 0 = bottom row). stats may be NULL. */
rtx_status rtx_multi_render(rtx_multi* m, const RtxCamera* cam, const RtxConfig* cfg, RtxFrame* out, RtxMultiStats* stats);
/* Convenience: create on devices 0..n_gpus-1 (block_rows 1), render once, destroy. */
rtx_status rtx_render_multi(const rtx_flat* f, const RtxCamera* cam, const RtxConfig* cfg, int32_t n_gpus, RtxFrame* out);

/* ---- the time-sweep renderer: render_scene_with_time(t0, t1, path, world)  world.rs:1249-1330 ------------------------ */
/* One frame of the reference's video experiment on a scene that is ALREADY resident on the GPU (many frames, one
 * upload): 500 x 500, 500 spp, depth 50, background (0.7, 0.8, 1), camera (13,2,3) -> (0,0,0), vfov 20, aspect 1,
 * aperture 0.1, focus 10, shutter [t0, t1) -- all hard-coded there -- written to `path` as P3 PPM.  The reference
 * renders it with its THREADS = 11 row bands (world.rs:18,1284), which leaves rows 495..499 black; row_chunk_compat = 1
 * reproduces that, 0 renders every row.  `overrides` may be NULL; a non-NULL RtxConfig replaces width / spp / depth /
 * seed (its aspect_ratio, background and threads are ignored) so that tests need not trace 125 M samples per frame. */
rtx_status rtx_render_scene_with_time(const rtx_scene* s, double t0, double t1, const char* path, int32_t row_chunk_compat,
                                      const RtxConfig* overrides);

/* ---- image output: Screen::write_to_ppm_file  screen.rs:40-59 ------------------------------ */
/* rgb8 in the row order above (row 0 = bottom); writes "P3\n{w} {h}\n255\n" then one "r g b" line
 * per pixel, top row first.  path NULL or "-" = stdout (Screen::write_to_ppm). */
rtx_status rtx_write_ppm(const char* path, int32_t width, int32_t height, const uint8_t* rgb8);

/* ---- device self-test -------------------------------------------------------------------------- */
/* Evaluates one arithmetic building block of the kernels ON THE GPU for n host-side operands
 * (copied in and out), so tests can prove it is bit-identical to the host's evaluation of the same
 * source.  fn: 0 sin, 1 cos, 2 log, 3 acos, 4 atan2(x,y), 5 tan, 6 sqrt, 7 x/y, 8 x*y+x (must NOT be
 * fused), 9 floor, 10 the sort key of a 4-wide BVH step for entry distance x and t_min y
 * (clamp into [y, 3e38] with NaN -> y).  rtx_device_stream: the first n uniforms of the (seed, pixel, sample) stream. */
rtx_status rtx_device_math(int32_t fn, const double* x, const double* y, int64_t n, double* out);
rtx_status rtx_device_stream(uint64_t seed, uint64_t pixel, uint32_t sample, int32_t n, double* out);

/* Opaque pass-through for the CPU checkers under oracle/ (test infrastructure; not a render path). */
const void* rtx_builder_graph(const rtx_builder* b);
const void* rtx_flat_arrays(const rtx_flat* f);

#ifdef __cplusplus
}
#endif
#endif /* RTX_ABI_H */
