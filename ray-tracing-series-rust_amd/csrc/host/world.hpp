// C++ host-side mirror of the reference's scene-construction API, header-only over the C ABI
// (include/rtx_abi.h).  The reference's host language is Rust, which this image cannot build, so
// the compiled-language host lives here: same type names, same constructor argument orders
// (`X::new(...)` in Rust is `X::new_(scene, ...)` here -- `new` is a C++ keyword and the scene
// object replaces Rust's implicit heap), same panics turned into exceptions at THIS layer only
// (nothing throws across the C ABI).
//
//   reference                                           here
//   Arc::new(Box::new(Sphere::new(c, r, mat)))          Sphere::new_(s, c, r, mat)
//   HittableList::new(); list.add(obj)                  HittableList::new_(s); list.add(obj)
//   BvhNode::from_list(&list, t0, t1)                   BvhNode::from_list(list, t0, t1)
//   get_world_cam(id) -> (world, cam, background)       get_world_cam(s, id)
//   render_scene(world, cam, background, config)        render_scene(s, world, cam, background, config)
#pragma once
#include <stdexcept>
#include <string>
#include <vector>
#include "../../../include/rtx_abi.h"

namespace rtsr {

struct Vec3 {  // vec3.rs:6
  double x, y, z;
  static Vec3 new_(double x, double y, double z) { return Vec3{x, y, z}; }
};
typedef Vec3 Point3;
typedef Vec3 Color;

struct Error : std::runtime_error {
  rtx_status status;
  Error(rtx_status s, const std::string& m) : std::runtime_error(m), status(s) {}
};
inline void check(rtx_status s) {
  if (s != RTX_OK) throw Error(s, rtx_last_error());
}
inline rtx_handle checked(rtx_handle h) {
  if (h < 0) throw Error(RTX_EINVAL, rtx_last_error());
  return h;
}

// Owns the scene graph under construction (the reference allocates Arc<Box<dyn ..>> objects instead).
class Scene {
 public:
  explicit Scene(uint64_t scene_seed = 1) : b_(nullptr) { check(rtx_builder_create(scene_seed, &b_)); }
  ~Scene() { rtx_builder_destroy(b_); }
  Scene(const Scene&) = delete;
  Scene& operator=(const Scene&) = delete;
  rtx_builder* builder() const { return b_; }

 private:
  rtx_builder* b_;
};

struct Texture { Scene* s; rtx_handle h; };
struct Material { Scene* s; rtx_handle h; };
struct Hittable { Scene* s; rtx_handle h; };

// ---- textures (texture.rs)
struct SolidColor {
  static Texture new_(Scene& s, const Color& c) { double v[3] = {c.x, c.y, c.z}; return {&s, checked(rtx_solid_color(s.builder(), v))}; }
  static Texture from_colors(Scene& s, double r, double g, double b) { return new_(s, Color{r, g, b}); }
};
struct Checker {
  static Texture new_(Scene& s, Texture even, Texture odd) { return {&s, checked(rtx_checker(s.builder(), even.h, odd.h))}; }
  static Texture from_colors(Scene& s, const Color& even, const Color& odd) { return new_(s, SolidColor::new_(s, even), SolidColor::new_(s, odd)); }
};
struct Noise {
  static Texture new_(Scene& s, double scale) { return {&s, checked(rtx_noise(s.builder(), scale))}; }
};
struct Image {
  static Texture from_ppm(Scene& s, const char* name) { return {&s, checked(rtx_image_from_ppm(s.builder(), name))}; }
};

// ---- materials (hit.rs:992-1152)
struct Lambertian {
  static Material new_(Scene& s, const Color& albedo) { return from_pointer(s, SolidColor::new_(s, albedo)); }
  static Material from_pointer(Scene& s, Texture t) { return {&s, checked(rtx_lambertian(s.builder(), t.h))}; }
};
struct Metal {
  static Material new_(Scene& s, const Color& albedo, double fuzz) { double v[3] = {albedo.x, albedo.y, albedo.z}; return {&s, checked(rtx_metal(s.builder(), v, fuzz))}; }
};
struct Dielectric {
  static Material new_(Scene& s, double ir) { return {&s, checked(rtx_dielectric(s.builder(), ir))}; }
};
struct DiffuseLight {
  static Material new_(Scene& s, const Color& c) { return from_pointer(s, SolidColor::new_(s, c)); }
  static Material from_pointer(Scene& s, Texture t) { return {&s, checked(rtx_diffuse_light(s.builder(), t.h))}; }
};
struct Isotropic {
  static Material from_color(Scene& s, const Color& c) { return {&s, checked(rtx_isotropic(s.builder(), SolidColor::new_(s, c).h))}; }
};

// ---- hittables (hit.rs, bvh.rs, model.rs)
struct Sphere {
  static Hittable new_(Scene& s, const Point3& center, double radius, Material m) {
    double c[3] = {center.x, center.y, center.z};
    return {&s, checked(rtx_sphere(s.builder(), c, radius, m.h))};
  }
};
struct MovingSphere {
  static Hittable new_(Scene& s, const Point3& c0, const Point3& c1, double t0, double t1, double radius, Material m) {
    double a[3] = {c0.x, c0.y, c0.z}, b[3] = {c1.x, c1.y, c1.z};
    return {&s, checked(rtx_moving_sphere(s.builder(), a, b, t0, t1, radius, m.h))};
  }
};
struct GravitySphere {  // hit.rs:340-367
  static Hittable new_(Scene& s, const Point3& start, double time0, double radius, Material m) {
    double a[3] = {start.x, start.y, start.z};
    return {&s, checked(rtx_gravity_sphere(s.builder(), a, time0, radius, m.h))};
  }
};
struct Triangle {
  static Hittable new_(Scene& s, const Point3& v0, const Point3& v1, const Point3& v2, Material m) {
    double a[3] = {v0.x, v0.y, v0.z}, b[3] = {v1.x, v1.y, v1.z}, c[3] = {v2.x, v2.y, v2.z};
    return {&s, checked(rtx_triangle(s.builder(), a, b, c, m.h))};
  }
};
struct XyRect {
  static Hittable new_(Scene& s, double x0, double x1, double y0, double y1, double k, Material m) { return {&s, checked(rtx_xy_rect(s.builder(), x0, x1, y0, y1, k, m.h))}; }
};
struct XzRect {
  static Hittable new_(Scene& s, double x0, double x1, double y0, double y1, double k, Material m) { return {&s, checked(rtx_xz_rect(s.builder(), x0, x1, y0, y1, k, m.h))}; }
};
struct YzRect {
  static Hittable new_(Scene& s, double x0, double x1, double y0, double y1, double k, Material m) { return {&s, checked(rtx_yz_rect(s.builder(), x0, x1, y0, y1, k, m.h))}; }
};
struct RectPrism {
  static Hittable new_(Scene& s, const Point3& p0, const Point3& p1, Material m) {
    double a[3] = {p0.x, p0.y, p0.z}, b[3] = {p1.x, p1.y, p1.z};
    return {&s, checked(rtx_rect_prism(s.builder(), a, b, m.h))};
  }
};
struct HittableList : Hittable {
  static HittableList new_(Scene& s) { HittableList l; l.s = &s; l.h = checked(rtx_hittable_list_new(s.builder())); return l; }
  void add(Hittable obj) { check(rtx_hittable_list_add(s->builder(), h, obj.h)); }
};
struct BvhNode {
  static Hittable from_list(const HittableList& list, double time0, double time1) {
    return {list.s, checked(rtx_bvh_from_list(list.s->builder(), list.h, time0, time1))};
  }
};
struct Translate {
  static Hittable new_(Scene& s, const Vec3& offset, Hittable obj) { double o[3] = {offset.x, offset.y, offset.z}; return {&s, checked(rtx_translate(s.builder(), o, obj.h))}; }
};
struct RotateY {
  static Hittable new_(Scene& s, double angle, Hittable obj) { return {&s, checked(rtx_rotate_y(s.builder(), angle, obj.h))}; }
};
struct ConstantMedium {
  static Hittable from_color(Scene& s, const Color& c, double density, Hittable boundary) {
    double v[3] = {c.x, c.y, c.z};
    return {&s, checked(rtx_constant_medium(s.builder(), v, density, boundary.h))};
  }
};
struct TriangleModel {  // model.rs:13-76: load_from_file(path, scale).to_hittable()
  static HittableList load_from_file(Scene& s, const char* path, double scale) {
    HittableList l; l.s = &s; l.h = checked(rtx_triangle_model(s.builder(), path, scale)); return l;
  }
};

// ---- camera / config / screen
struct Camera {  // camera.rs:20-57
  RtxCamera c;
  static Camera new_(const Point3& lookfrom, const Point3& lookat, const Vec3& vup, double vfov, double aspect_ratio,
                     double aperture, double focus_dist, double time1, double time2) {
    Camera cam;
    double a[3] = {lookfrom.x, lookfrom.y, lookfrom.z}, b[3] = {lookat.x, lookat.y, lookat.z}, u[3] = {vup.x, vup.y, vup.z};
    check(rtx_camera_new(a, b, u, vfov, aspect_ratio, aperture, focus_dist, time1, time2, &cam.c));
    return cam;
  }
};
struct Config {  // world.rs:20-50
  RtxConfig c;
  static Config new_(double aspect_ratio, int image_width, int samples_per_pixel, int max_depth, size_t threads) {
    Config cfg;
    check(rtx_config_new(aspect_ratio, image_width, samples_per_pixel, max_depth, (int32_t)threads, &cfg.c));
    return cfg;
  }
};
struct Screen {  // screen.rs:6-59, row 0 = bottom row
  int width = 0, height = 0;
  std::vector<uint8_t> rgb8;
  std::vector<double> accum;
  void write_to_ppm() const { check(rtx_write_ppm(nullptr, width, height, rgb8.data())); }
  void write_to_ppm_file(const char* path) const { check(rtx_write_ppm(path, width, height, rgb8.data())); }
};

struct WorldCam {
  Hittable world;
  Camera cam;
  Color background;
};
// get_world_cam(config_num) -- world.rs:876-1179
inline WorldCam get_world_cam(Scene& s, int config_num, const RtxSceneOptions* options = nullptr) {
  WorldCam wc;
  double bg[3];
  rtx_handle h = -1;
  check(rtx_get_world_cam(s.builder(), config_num, options, &h, &wc.cam.c, bg));
  wc.world = Hittable{&s, h};
  wc.background = Color{bg[0], bg[1], bg[2]};
  return wc;
}

// render_scene(world, cam, background, config) -- world.rs:1181-1247 -- on the current GPU.
// Returns the Screen; the reference prints it (call write_to_ppm()).
inline Screen render_scene(Scene& s, Hittable world, const Camera& cam, const Color& background, Config config,
                           RtxRenderStats* stats = nullptr) {
  config.c.background[0] = background.x; config.c.background[1] = background.y; config.c.background[2] = background.z;
  rtx_flat* flat = nullptr;
  check(rtx_flatten(s.builder(), world.h, nullptr, &flat));
  rtx_scene* scene = nullptr;
  rtx_status st = rtx_scene_upload(flat, &scene);
  rtx_flat_destroy(flat);
  check(st);
  Screen scr;
  scr.width = config.c.image_width;
  scr.height = rtx_image_height(&config.c);
  scr.rgb8.resize((size_t)scr.width * scr.height * 3);
  scr.accum.resize((size_t)scr.width * scr.height * 3);
  RtxFrame frame = {scr.accum.data(), scr.rgb8.data()};
  st = rtx_render(scene, &cam.c, &config.c, &frame);
  (void)stats;
  rtx_scene_destroy(scene);
  check(st);
  return scr;
}

// render_scene on n GPUs of this process: row-interleaved shards, one RCCL gather (replaces the band threads and the
// collect loop of world.rs:1198-1244 at node scale).
inline Screen render_scene_multi(Scene& s, Hittable world, const Camera& cam, const Color& background, Config config, int n_gpus) {
  config.c.background[0] = background.x; config.c.background[1] = background.y; config.c.background[2] = background.z;
  rtx_flat* flat = nullptr;
  check(rtx_flatten(s.builder(), world.h, nullptr, &flat));
  Screen scr;
  scr.width = config.c.image_width;
  scr.height = rtx_image_height(&config.c);
  scr.rgb8.resize((size_t)scr.width * scr.height * 3);
  scr.accum.resize((size_t)scr.width * scr.height * 3);
  RtxFrame frame = {scr.accum.data(), scr.rgb8.data()};
  rtx_status st = rtx_render_multi(flat, &cam.c, &config.c, n_gpus, &frame);
  rtx_flat_destroy(flat);
  check(st);
  return scr;
}

// render_scene_with_time(t0, t1, path, world) -- world.rs:1249-1330.  A ResidentWorld keeps the scene on the GPU across
// frames (the reference's video loop rebuilt nothing either: it shares one Arc<world>).
struct ResidentWorld {
  rtx_scene* scene = nullptr;
  ResidentWorld(Scene& s, Hittable world) {
    rtx_flat* flat = nullptr;
    check(rtx_flatten(s.builder(), world.h, nullptr, &flat));
    rtx_status st = rtx_scene_upload(flat, &scene);
    rtx_flat_destroy(flat);
    check(st);
  }
  ~ResidentWorld() { rtx_scene_destroy(scene); }
  ResidentWorld(const ResidentWorld&) = delete;
  ResidentWorld& operator=(const ResidentWorld&) = delete;
};
inline void render_scene_with_time(double t0, double t1, const char* path, const ResidentWorld& world, bool row_chunk_compat = true) {
  check(rtx_render_scene_with_time(world.scene, t0, t1, path, row_chunk_compat ? 1 : 0, nullptr));
}

}  // namespace rtsr
