// Host-only entry points of include/rtx_abi.h: builder (one call per reference constructor),
// camera/config, scene catalogue, flatten, PPM output.  Device entry points: render.hip.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include "../host/scenes.hpp"
#include "abi_internal.hpp"

namespace rtx {
static thread_local std::string g_last_error;
void set_error(const std::string& msg) { g_last_error = msg; }
}  // namespace rtx

using namespace rtx;

namespace {
inline rtx_handle checked(rtx_builder* b, int32_t h) {
  if (h < 0) set_error(b->graph.error.empty() ? "invalid argument" : b->graph.error);
  return h;
}
#define NEED_BUILDER(b)                                  \
  if (!(b)) { set_error("NULL builder"); return -1; }
}  // namespace

extern "C" {

int32_t rtx_abi_version(void) { return RTX_ABI_VERSION; }
const char* rtx_last_error(void) { return g_last_error.c_str(); }
const char* rtx_trace_kernel_name(int32_t kernel) {
  switch (kernel) {
    case RTX_KERNEL_SIMPLE: return "k_trace_simple";
    case RTX_KERNEL_PERSISTENT: return "k_trace_persistent";
    case RTX_KERNEL_STREAM: return "k_trace_stream";
    case RTX_KERNEL_VOTE: return "k_trace_vote";
    case RTX_KERNEL_LDS: return "k_trace_lds";
    case RTX_KERNEL_WQ: return "k_trace_wq";
    case RTX_KERNEL_WORLD: return "k_trace_world";
    case RTX_KERNEL_WAVEFRONT: return "k_wf_trace";
    default: return "?";
  }
}

rtx_status rtx_builder_create(uint64_t scene_seed, rtx_builder** out) {
  if (!out) { set_error("rtx_builder_create: NULL out"); return RTX_EINVAL; }
  *out = new (std::nothrow) rtx_builder(scene_seed);
  if (!*out) { set_error("out of memory"); return RTX_ENOMEM; }
  return RTX_OK;
}
void rtx_builder_destroy(rtx_builder* b) { delete b; }
double rtx_builder_random(rtx_builder* b) { return b ? rt::host_rng_f64(b->graph.rng) : 0.0; }

rtx_handle rtx_solid_color(rtx_builder* b, const double rgb[3]) { NEED_BUILDER(b); if (!rgb) { set_error("NULL rgb"); return -1; } return checked(b, b->graph.solid_color(rgb)); }
rtx_handle rtx_checker(rtx_builder* b, rtx_handle even, rtx_handle odd) { NEED_BUILDER(b); return checked(b, b->graph.checker(even, odd)); }
rtx_handle rtx_noise(rtx_builder* b, double scale) { NEED_BUILDER(b); return checked(b, b->graph.noise(scale)); }
rtx_handle rtx_image_from_ppm(rtx_builder* b, const char* path) { NEED_BUILDER(b); if (!path) { set_error("NULL path"); return -1; } return checked(b, b->graph.image_from_ppm(path)); }
rtx_handle rtx_image_from_texels(rtx_builder* b, int32_t w, int32_t h, const double* t) { NEED_BUILDER(b); return checked(b, b->graph.image_from_texels(w, h, t)); }

rtx_handle rtx_lambertian(rtx_builder* b, rtx_handle tex) { NEED_BUILDER(b); return checked(b, b->graph.lambertian(tex)); }
rtx_handle rtx_metal(rtx_builder* b, const double albedo[3], double fuzz) { NEED_BUILDER(b); if (!albedo) { set_error("NULL albedo"); return -1; } return checked(b, b->graph.metal(albedo, fuzz)); }
rtx_handle rtx_dielectric(rtx_builder* b, double ir) { NEED_BUILDER(b); return checked(b, b->graph.dielectric(ir)); }
rtx_handle rtx_diffuse_light(rtx_builder* b, rtx_handle tex) { NEED_BUILDER(b); return checked(b, b->graph.diffuse_light(tex)); }
rtx_handle rtx_isotropic(rtx_builder* b, rtx_handle tex) { NEED_BUILDER(b); return checked(b, b->graph.isotropic(tex)); }

rtx_handle rtx_sphere(rtx_builder* b, const double c[3], double r, rtx_handle mat) { NEED_BUILDER(b); if (!c) { set_error("NULL center"); return -1; } return checked(b, b->graph.sphere(c, r, mat)); }
rtx_handle rtx_gravity_sphere(rtx_builder* b, const double start[3], double time0, double radius, rtx_handle mat) {
  if (!b || !start) { set_error("rtx_gravity_sphere: NULL argument"); return -1; }
  return checked(b, b->graph.gravity_sphere(start, time0, radius, mat));
}
rtx_handle rtx_moving_sphere(rtx_builder* b, const double c0[3], const double c1[3], double t0, double t1, double r, rtx_handle mat) {
  NEED_BUILDER(b);
  if (!c0 || !c1) { set_error("NULL center"); return -1; }
  return checked(b, b->graph.moving_sphere(c0, c1, t0, t1, r, mat));
}
rtx_handle rtx_triangle(rtx_builder* b, const double v0[3], const double v1[3], const double v2[3], rtx_handle mat) {
  NEED_BUILDER(b);
  if (!v0 || !v1 || !v2) { set_error("NULL vertex"); return -1; }
  return checked(b, b->graph.triangle(v0, v1, v2, mat));
}
rtx_handle rtx_xy_rect(rtx_builder* b, double x0, double x1, double y0, double y1, double k, rtx_handle mat) { NEED_BUILDER(b); return checked(b, b->graph.rect(H_XY_RECT, x0, x1, y0, y1, k, mat)); }
rtx_handle rtx_xz_rect(rtx_builder* b, double x0, double x1, double y0, double y1, double k, rtx_handle mat) { NEED_BUILDER(b); return checked(b, b->graph.rect(H_XZ_RECT, x0, x1, y0, y1, k, mat)); }
rtx_handle rtx_yz_rect(rtx_builder* b, double x0, double x1, double y0, double y1, double k, rtx_handle mat) { NEED_BUILDER(b); return checked(b, b->graph.rect(H_YZ_RECT, x0, x1, y0, y1, k, mat)); }
rtx_handle rtx_rect_prism(rtx_builder* b, const double p0[3], const double p1[3], rtx_handle mat) {
  NEED_BUILDER(b);
  if (!p0 || !p1) { set_error("NULL corner"); return -1; }
  return checked(b, b->graph.rect_prism(p0, p1, mat));
}
rtx_handle rtx_hittable_list_new(rtx_builder* b) { NEED_BUILDER(b); return b->graph.list_new(); }
rtx_status rtx_hittable_list_add(rtx_builder* b, rtx_handle list, rtx_handle object) {
  if (!b) { set_error("NULL builder"); return RTX_EINVAL; }
  if (!b->graph.list_add(list, object)) { set_error(b->graph.error); return RTX_EINVAL; }
  return RTX_OK;
}
rtx_handle rtx_bvh_from_list(rtx_builder* b, rtx_handle list, double t0, double t1) { NEED_BUILDER(b); return checked(b, b->graph.bvh_from_list(list, t0, t1)); }
rtx_handle rtx_translate(rtx_builder* b, const double offset[3], rtx_handle obj) { NEED_BUILDER(b); if (!offset) { set_error("NULL offset"); return -1; } return checked(b, b->graph.translate(offset, obj)); }
rtx_handle rtx_rotate_y(rtx_builder* b, double angle, rtx_handle obj) { NEED_BUILDER(b); return checked(b, b->graph.rotate_y(angle, obj)); }
rtx_handle rtx_constant_medium(rtx_builder* b, const double rgb[3], double density, rtx_handle boundary) {
  NEED_BUILDER(b);
  if (!rgb) { set_error("NULL rgb"); return -1; }
  return checked(b, b->graph.constant_medium(rgb, density, boundary));
}
rtx_handle rtx_triangle_model(rtx_builder* b, const char* path, double scale) { NEED_BUILDER(b); if (!path) { set_error("NULL path"); return -1; } return checked(b, b->graph.triangle_model(path, scale)); }
rtx_handle rtx_triangle_mesh(rtx_builder* b, const double* vertices, int64_t n_vertices, const int64_t* faces, int64_t n_faces, rtx_handle mat) {
  NEED_BUILDER(b);
  if (!vertices || !faces || n_vertices < 0 || n_faces < 0) { set_error("triangle_mesh: NULL or negative"); return -1; }
  return checked(b, b->graph.triangle_mesh(vertices, n_vertices, faces, n_faces, mat));
}

rtx_status rtx_camera_new(const double lookfrom[3], const double lookat[3], const double vup[3],
                          double vfov, double aspect_ratio, double aperture, double focus_dist,
                          double time1, double time2, RtxCamera* out) {
  if (!lookfrom || !lookat || !vup || !out) { set_error("rtx_camera_new: NULL argument"); return RTX_EINVAL; }
  if (!(time1 < time2)) { set_error("rtx_camera_new: time1 >= time2 (gen_range(time1..time2) panics, camera.rs:69)"); return RTX_EINVAL; }
  rt::FlatCamera c = camera_new(lookfrom, lookat, vup, vfov, aspect_ratio, aperture, focus_dist, time1, time2);
  static_assert(sizeof(RtxCamera) == sizeof(rt::FlatCamera), "camera layout");
  memcpy(out, &c, sizeof(c));
  return RTX_OK;
}

rtx_status rtx_config_new(double aspect_ratio, int32_t image_width, int32_t samples_per_pixel,
                          int32_t max_depth, int32_t threads, RtxConfig* out) {
  if (!out) { set_error("rtx_config_new: NULL out"); return RTX_EINVAL; }
  // world.rs:36-40
  if (threads <= 0) { set_error("Config::new: assert!(threads > 0)"); return RTX_EINVAL; }
  if (image_width <= 0) { set_error("Config::new: assert!(image_width > 0)"); return RTX_EINVAL; }
  if (samples_per_pixel <= 0) { set_error("Config::new: assert!(samples_per_pixel > 0)"); return RTX_EINVAL; }
  if (max_depth <= 0) { set_error("Config::new: assert!(max_depth > 0)"); return RTX_EINVAL; }
  memset(out, 0, sizeof(*out));
  out->aspect_ratio = aspect_ratio;
  out->image_width = image_width;
  out->samples_per_pixel = samples_per_pixel;
  out->max_depth = max_depth;
  out->threads = threads;
  out->seed = 1;
  out->background[0] = 0.7; out->background[1] = 0.8; out->background[2] = 1.0;
  return RTX_OK;
}

int32_t rtx_image_height(const RtxConfig* cfg) {
  if (!cfg) return 0;
  return rt::rt_f64_as_i32((double)cfg->image_width / cfg->aspect_ratio);  // world.rs:1192
}

int32_t rtx_shard_rows(const RtxConfig* cfg, const RtxShard* shard) {
  if (!cfg) return 0;
  int32_t h = rtx_image_height(cfg);
  RtxShard sh = {0, 1, 1, 0};
  if (shard) sh = *shard;
  if (sh.shard_count <= 0 || sh.block_rows <= 0) return 0;
  int n = 0;
  for (int32_t j = 0; j < h; ++j)
    if ((j / sh.block_rows) % sh.shard_count == sh.shard_index) ++n;
  return n;
}

rtx_status rtx_get_world_cam(rtx_builder* b, int32_t scene_id, const RtxSceneOptions* options,
                             rtx_handle* world_out, RtxCamera* cam_out, double background_out[3]) {
  if (!b || !world_out || !cam_out || !background_out) { set_error("rtx_get_world_cam: NULL argument"); return RTX_EINVAL; }
  SceneOptions opt;
  if (options) {
    opt.camera_aspect = options->camera_aspect;
    opt.earth_ppm = options->earth_ppm;
    opt.dragon_ply = options->dragon_ply;
    if (options->mesh_triangles > 0) opt.mesh_triangles = options->mesh_triangles;
    if (options->book2_boxes_per_side > 0) opt.book2_boxes_per_side = options->book2_boxes_per_side;
    if (options->book2_spheres > 0) opt.book2_spheres = options->book2_spheres;
  }
  WorldCam wc;
  std::string err;
  if (!get_world_cam(b->graph, scene_id, opt, &wc, &err)) {
    set_error(err);
    return scene_id == RTX_SCENE_RANDOM_MOVING ? RTX_EUNSUPPORTED : RTX_EINVAL;
  }
  *world_out = wc.world;
  memcpy(cam_out, &wc.cam, sizeof(wc.cam));
  background_out[0] = wc.background[0]; background_out[1] = wc.background[1]; background_out[2] = wc.background[2];
  return RTX_OK;
}

rtx_status rtx_flatten(const rtx_builder* b, rtx_handle world, const RtxBuildOptions* options, rtx_flat** out) {
  if (!b || !out) { set_error("rtx_flatten: NULL argument"); return RTX_EINVAL; }
  *out = nullptr;
  BuildOptions opt;
  if (options) {
    if (options->max_leaf > 0) opt.max_leaf = options->max_leaf;
    if (options->sah_bins > 0) opt.sah_bins = options->sah_bins;
    opt.reference_bvh = options->reference_bvh ? 1 : 0;
    opt.gpu_builder = options->gpu_builder ? 1 : 0;
    if (options->bvh_seed) opt.bvh_seed = options->bvh_seed;
  }
  if (const char* e = getenv("RTX_LEAF_FIRST")) opt.leaf_first = atoi(e) != 0 ? 1 : 0;  // A/B switches, never change a result
  if (const char* e = getenv("RTX_MOTION_TOPOLOGY")) opt.motion_topology = atoi(e) != 0 ? 1 : 0;
  rtx_flat* f = new (std::nothrow) rtx_flat();
  if (!f) { set_error("out of memory"); return RTX_ENOMEM; }
  std::string err;
  if (!b->graph.valid_hittable(world)) { delete f; set_error("rtx_flatten: bad world handle"); return RTX_EINVAL; }
  if (!flatten_scene(b->graph, world, opt, &f->scene, &err)) {
    delete f;
    set_error(err);
    return RTX_EUNSUPPORTED;
  }
  *out = f;
  return RTX_OK;
}
void rtx_flat_destroy(rtx_flat* f) { delete f; }

rtx_status rtx_flat_info(const rtx_flat* f, RtxFlatInfo* o) {
  if (!f || !o) { set_error("rtx_flat_info: NULL argument"); return RTX_EINVAL; }
  const FlatScene& s = f->scene;
  o->n_spheres = (int64_t)s.spheres.size(); o->n_moving_spheres = (int64_t)s.moving_spheres.size();
  o->n_rects = (int64_t)s.rects.size(); o->n_triangles = (int64_t)s.triangles.size();
  o->n_nodes = (int64_t)s.nodes.size(); o->n_refs = (int64_t)s.refs.size();
  o->n_entries = (int64_t)s.entries.size(); o->n_top_level = (int64_t)s.top_level.size();
  o->n_materials = (int64_t)s.materials.size(); o->n_textures = (int64_t)s.textures.size();
  o->n_perlins = (int64_t)s.perlins.size(); o->n_images = (int64_t)s.images.size();
  o->n_texels = (int64_t)(s.texels.size() / 3);
  o->total_bytes = (int64_t)s.total_bytes();
  o->max_stack = s.max_stack; o->n_bvh = s.n_bvh; o->sah_cost = s.sah_cost;
  o->bvh_build_ms = s.bvh_build_ms; o->bvh_device_ms = s.bvh_device_ms;
  o->n_gravity_spheres = (int64_t)s.gravity_spheres.size();
  return RTX_OK;
}

int32_t rtx_flat_top_level_kind(const rtx_flat* f, int32_t index) {
  if (!f || index < 0 || (size_t)index >= f->scene.top_level.size()) return -1;
  return f->scene.entries[f->scene.top_level[index]].kind;
}

// screen.rs:40-59 + vec3.rs:109-114: integer-valued channels printed without a decimal point.
rtx_status rtx_write_ppm(const char* path, int32_t width, int32_t height, const uint8_t* rgb8) {
  if (width <= 0 || height <= 0 || !rgb8) { set_error("rtx_write_ppm: bad argument"); return RTX_EINVAL; }
  FILE* fp = stdout;
  bool to_file = path && strcmp(path, "-") != 0;
  if (to_file) {
    fp = fopen(path, "wb");
    if (!fp) { set_error(std::string("rtx_write_ppm: cannot open ") + path); return RTX_EIO; }
  }
  std::string buf;
  buf.reserve((size_t)width * height * 12 + 32);
  char line[64];
  snprintf(line, sizeof(line), "P3\n%d %d\n255\n", width, height);
  buf += line;
  for (int32_t j = height - 1; j >= 0; --j) {
    for (int32_t i = 0; i < width; ++i) {
      const uint8_t* p = rgb8 + 3 * ((size_t)j * width + i);
      int n = snprintf(line, sizeof(line), "%u %u %u\n", (unsigned)p[0], (unsigned)p[1], (unsigned)p[2]);
      buf.append(line, (size_t)n);
    }
  }
  size_t wrote = fwrite(buf.data(), 1, buf.size(), fp);
  if (to_file) fclose(fp); else fflush(fp);
  if (wrote != buf.size()) { set_error("rtx_write_ppm: short write"); return RTX_EIO; }
  return RTX_OK;
}

const void* rtx_builder_graph(const rtx_builder* b) { return b ? (const void*)&b->graph : nullptr; }
const void* rtx_flat_arrays(const rtx_flat* f) { return f ? (const void*)&f->scene : nullptr; }

}  // extern "C"
