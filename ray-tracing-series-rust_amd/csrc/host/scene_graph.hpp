// The scene as the reference's constructors describe it: an object graph of
// Hittables, Materials and Textures, kept as plain records (kind + numbers +
// child handles).  It is what `rtx_builder` holds behind the C ABI.
//
// It mirrors the constructor surface of /root/reference/src/hit.rs,
// texture.rs, perlin.rs and model.rs one to one (argument order included) and
// holds NO intersection or shading code: the product flattens it
// (host/flatten.cpp) for the HIP kernels; the literal CPU checker under
// oracle/ interprets it independently.
#pragma once
#include <stdint.h>
#include <string>
#include <vector>
#include "../core/flat_types.hpp"
#include "../core/rng.hpp"

namespace rtx {

enum HittableKind : int32_t {
  H_SPHERE = 0,          // Sphere::new(center, radius, mat)                    hit.rs:187-193
  H_MOVING_SPHERE = 1,   // MovingSphere::new(c0, c1, t0, t1, r, mat)           hit.rs:257-273
  H_TRIANGLE = 2,        // Triangle::new(v0, v1, v2, mat)                      hit.rs:96-107
  H_XY_RECT = 3,         // XyRect::new(x0, x1, y0, y1, k, mat)                 hit.rs:456-472
  H_XZ_RECT = 4,         // XzRect::new(...)                                    hit.rs:521-537
  H_YZ_RECT = 5,         // YzRect::new(...)                                    hit.rs:586-602
  H_LIST = 6,            // HittableList::new + add                             hit.rs:646-652
  H_RECT_PRISM = 7,      // RectPrism::new(p0, p1, mat)                         hit.rs:720-775
  H_BVH = 8,             // BvhNode::from_list(list, time0, time1)              bvh.rs:85-93
  H_TRANSLATE = 9,       // Translate::new(offset, obj)                         hit.rs:793-798
  H_ROTATE_Y = 10,       // RotateY::new(angle_deg, obj)                        hit.rs:843-888
  H_CONSTANT_MEDIUM = 11,// ConstantMedium::from_color(color, density, boundary) hit.rs:945-951
  H_GRAVITY_SPHERE = 12  // GravitySphere::new(start, time0, radius, mat)         hit.rs:340-367
};

struct GHittable {
  int32_t kind;
  int32_t mat;                    // material handle (-1 if none)
  double f[12];                   // constructor numbers, kind specific (see scene_graph.cpp)
  std::vector<int32_t> children;  // LIST/BVH: objects in add() order; wrappers: the one child
  std::vector<double> table;      // GRAVITY_SPHERE: `stored`, the simulated heights (hit.rs:346-359)
};

struct GMaterial {
  int32_t kind;  // rt::MaterialKind
  int32_t tex;
  double albedo[3];
  double param;
};

struct GTexture {
  int32_t kind;  // rt::TextureKind
  int32_t a, b;
  double color[3];
  double scale;
};

struct GImage {
  int32_t width, height;
  std::vector<double> texels;  // 3 per pixel, row-major as Screen stores them (screen.rs:75-88)
};

struct SceneGraph {
  std::vector<GHittable> hittables;
  std::vector<GMaterial> materials;
  std::vector<GTexture> textures;
  std::vector<rt::FlatPerlin> perlins;
  std::vector<GImage> images;
  rt::HostRng rng;  // scene-construction stream (the reference uses thread_rng() here too)
  std::string error;

  explicit SceneGraph(uint64_t scene_seed) { rng.state = scene_seed; }

  // --- textures (texture.rs) ---
  int32_t solid_color(const double rgb[3]);
  int32_t checker(int32_t even, int32_t odd);
  int32_t noise(double scale);  // Noise::new -> Perlin::new draws the tables from `rng`
  int32_t image_from_texels(int32_t w, int32_t h, const double* texels);
  int32_t image_from_ppm(const char* path);  // Image::from_ppm -> Screen::from_ppm_p3
  // --- materials (hit.rs:992-1152) ---
  int32_t lambertian(int32_t tex);
  int32_t metal(const double albedo[3], double fuzz);
  int32_t dielectric(double ir);
  int32_t diffuse_light(int32_t tex);
  int32_t isotropic(int32_t tex);
  // --- hittables ---
  int32_t sphere(const double c[3], double radius, int32_t mat);
  int32_t moving_sphere(const double c0[3], const double c1[3], double t0, double t1, double radius, int32_t mat);
  int32_t gravity_sphere(const double start[3], double time0, double radius, int32_t mat);
  int32_t triangle(const double v0[3], const double v1[3], const double v2[3], int32_t mat);
  int32_t rect(int32_t kind, double a0, double a1, double b0, double b1, double k, int32_t mat);
  int32_t rect_prism(const double p0[3], const double p1[3], int32_t mat);
  int32_t list_new();
  bool list_add(int32_t list, int32_t obj);
  int32_t bvh_from_list(int32_t list, double time0, double time1);
  int32_t translate(const double offset[3], int32_t obj);
  int32_t rotate_y(double angle_deg, int32_t obj);
  int32_t constant_medium(const double rgb[3], double density, int32_t boundary);
  // model.rs:13-76: ASCII PLY -> list of triangles, one Lambertian(0.2,0.2,0.2) shared
  // (the reference allocates an identical one per face).
  int32_t triangle_model(const char* path, double scale);
  int32_t triangle_mesh(const double* vertices, int64_t n_vertices, const int64_t* faces,
                        int64_t n_faces, int32_t mat);

  bool valid_hittable(int32_t h) const { return h >= 0 && (size_t)h < hittables.size(); }
  bool valid_material(int32_t m) const { return m >= 0 && (size_t)m < materials.size(); }
  bool valid_texture(int32_t t) const { return t >= 0 && (size_t)t < textures.size(); }
};

// Perlin::new (perlin.rs:14-26, 68-83): 256 gradient vectors in [-1,1)^3, then three
// permutations shuffled for i = 254 .. 1 (index 255 is never touched).
void perlin_generate(rt::HostRng& rng, rt::FlatPerlin* out);

// Screen::from_ppm_p3 (screen.rs:61-95): P3 text, no comments, skips exactly three
// header lines' worth (magic, "w h", maxval).  Returns false and sets *err on I/O trouble.
bool read_ppm_p3(const char* path, int32_t* w, int32_t* h, std::vector<double>* texels,
                 std::string* err);

}  // namespace rtx
