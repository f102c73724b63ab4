// ORACLE O1 -- TEST INFRASTRUCTURE, NOT A RENDER PATH OF THE PRODUCT.
//
// A literal CPU restatement of the reference's hot path: the same object graph
// (trait objects -> virtual classes), the same recursive BvhNode with the reference's
// build rule, HittableList linear scans, per-candidate HitRecords, and render_scene's
// row-band threads.  It shares NO intersection/shading/vector code with the product
// (ray-tracing-series-rust_amd/csrc/core): Vec3, Aabb, every hit(), scatter(), value() and
// the Perlin code below are written from the reference source alone.  Shared on purpose are
// only (a) the scene graph records (plain data the product's builder collects), (b) the
// counter-based RNG (core/rng.hpp) that replaces rand::thread_rng(), and (c) core/rt_math.hpp,
// the deterministic libm both sides are specified to use.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call this.
//
// Parity status: the reference (Rust, rand 0.8.5 OS-seeded) can be neither built nor run
// here and its own tests (src/vec3.rs:343-428) pin only Vec3 algebra, so for the hot path
// this oracle is pinned by (1) those 9 Vec3 tests, (2) hand-derived known-answer tests
//  and (3) the three images the reference ships (images/book1.png, book2.png,
// stanford_dragon.png): silhouettes / wall edges / the light's outline to a pixel or two and the mean radiance
// of regions its unseeded randomness does not shape to a few per cent (tests/test_reference_image_pins.py,
// fixture + generator under tests/golden/).  No output of the reference can be reproduced bit for bit
// (OS-seeded thread_rng): at the level of bits this oracle remains "parity unpinned".
//
// Each function cites the reference lines it follows (paths relative to /root/reference/src).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <atomic>
#include <thread>
#include <unordered_map>
#include <vector>
#include "../ray-tracing-series-rust_amd/csrc/core/rng.hpp"
#include "../ray-tracing-series-rust_amd/csrc/core/rt_math.hpp"
#include "../ray-tracing-series-rust_amd/csrc/host/scene_graph.hpp"
#include "oracle_abi.h"

namespace o1 {

using rt::Rng;

// ------------------------------------------------------------------ vec3.rs
struct Vec3 {
  double e0, e1, e2;  // vec3.rs:6
  Vec3() : e0(0), e1(0), e2(0) {}
  Vec3(double x, double y, double z) : e0(x), e1(y), e2(z) {}
  double x() const { return e0; }
  double y() const { return e1; }
  double z() const { return e2; }
  double dot(const Vec3& o) const { return x() * o.x() + y() * o.y() + z() * o.z(); }  // vec3.rs:43-45
  double length_squared() const { return dot(*this); }                                 // vec3.rs:35-37
  double length() const { return std::sqrt(length_squared()); }                        // vec3.rs:39-41
  Vec3 cross(const Vec3& o) const {                                                    // vec3.rs:47-53
    return Vec3(y() * o.z() - z() * o.y(), z() * o.x() - x() * o.z(), x() * o.y() - y() * o.x());
  }
  bool near_zero() const {  // vec3.rs:59-62
    double s = 1e-8;
    return std::fabs(x()) < s && std::fabs(y()) < s && std::fabs(z()) < s;
  }
};
typedef Vec3 Point3;
typedef Vec3 Color;
static Vec3 operator*(const Vec3& a, const Vec3& b) { return Vec3(a.x() * b.x(), a.y() * b.y(), a.z() * b.z()); }  // 139-149
static Vec3 operator-(const Vec3& a) { return Vec3(a.x() * -1.0, a.y() * -1.0, a.z() * -1.0); }                   // 151-161
static Vec3 operator*(const Vec3& a, double t) { return Vec3(a.x() * t, a.y() * t, a.z() * t); }                  // 163-173
static Vec3 operator*(double t, const Vec3& a) { return Vec3(t * a.x(), t * a.y(), t * a.z()); }                  // 175-185
static Vec3 operator/(const Vec3& a, double t) { return Vec3(a.x() / t, a.y() / t, a.z() / t); }                  // 187-197
static Vec3 operator+(const Vec3& a, const Vec3& b) { return Vec3(a.x() + b.x(), a.y() + b.y(), a.z() + b.z()); } // 199-209
static Vec3 operator-(const Vec3& a, const Vec3& b) { return Vec3(a.x() - b.x(), a.y() - b.y(), a.z() - b.z()); } // 211-221
static void operator+=(Vec3& a, const Vec3& b) { a.e0 += b.x(); a.e1 += b.y(); a.e2 += b.z(); }                    // 223-229
static void operator*=(Vec3& a, const Vec3& b) { a.e0 *= b.x(); a.e1 *= b.y(); a.e2 *= b.z(); }                    // 245-251
static Vec3 unit(const Vec3& a) { return a / a.length(); }                                                          // 55-57
static Vec3 reflect(const Vec3& v, const Vec3& n) { return v - 2.0 * v.dot(n) * n; }                                // 64-66
static Vec3 refract(const Vec3& uv, const Vec3& n, double etai_over_etat) {                                         // 116-121
  double cos_theta = std::fmin((-uv).dot(n), 1.0);
  Vec3 r_out_perp = etai_over_etat * (uv + cos_theta * n);
  Vec3 r_out_parallel = -(std::sqrt(std::fabs(1.0 - r_out_perp.length_squared()))) * n;
  return r_out_perp + r_out_parallel;
}
static double clamp(double x, double mn, double mx) { return x < mn ? mn : (x > mx ? mx : x); }  // mutil.rs:1-9
static int32_t as_i32(double x) {  // Rust `as i32`: saturating, NaN -> 0
  if (x != x) return 0;
  if (x >= 2147483647.0) return 2147483647;
  if (x <= -2147483648.0) return (int32_t)0x80000000u;
  return (int32_t)x;
}
// vec3.rs:89-107
static void get_normalized_color(const Color& c, uint32_t spp, int32_t out[3]) {
  const double COLOR_MAX = 255.9;
  double r = c.x(), g = c.y(), b = c.z();
  double scale = 1.0 / (double)spp;
  r *= scale; g *= scale; b *= scale;
  r = std::sqrt(r); g = std::sqrt(g); b = std::sqrt(b);
  out[0] = as_i32(COLOR_MAX * clamp(r, 0.0, 1.0));
  out[1] = as_i32(COLOR_MAX * clamp(g, 0.0, 1.0));
  out[2] = as_i32(COLOR_MAX * clamp(b, 0.0, 1.0));
}
// vec3.rs:273-322, drawing from the path's counter-based stream instead of thread_rng()
static Vec3 random_range(Rng& g, double mn, double mx) {
  double x = rt::rng_range(g, mn, mx);
  double y = rt::rng_range(g, mn, mx);
  double z = rt::rng_range(g, mn, mx);
  return Vec3(x, y, z);
}
static Vec3 random_in_unit_sphere(Rng& g) {
  for (;;) {
    Vec3 p = random_range(g, -1.0, 1.0);
    if (p.length_squared() < 1.0) return p;
  }
}
static Vec3 random_unit_vector(Rng& g) { return unit(random_in_unit_sphere(g)); }
static Vec3 random_in_unit_disk(Rng& g) {
  for (;;) {
    double x = rt::rng_range(g, -1.0, 1.0);
    double y = rt::rng_range(g, -1.0, 1.0);
    Vec3 p(x, y, 0);
    if (p.length_squared() < 1.0) return p;
  }
}

// ------------------------------------------------------------------ ray.rs
struct Ray {
  Point3 origin;
  Vec3 direction;
  double time;
  Ray() : time(0) {}
  Ray(const Point3& o, const Vec3& d, double t) : origin(o), direction(d), time(t) {}
  Point3 at(double t) const { return origin + direction * t; }  // ray.rs:31-33
};

// ------------------------------------------------------------------ aabb.rs
struct Aabb {
  Point3 minimum, maximum;
  Aabb() {}
  Aabb(const Point3& a, const Point3& b) : minimum(a), maximum(b) {}
  bool hit(const Ray& r, double t_min, double t_max) const {  // aabb.rs:23-61
    const double mins[3] = {minimum.x(), minimum.y(), minimum.z()};
    const double maxs[3] = {maximum.x(), maximum.y(), maximum.z()};
    const double orig[3] = {r.origin.x(), r.origin.y(), r.origin.z()};
    const double dirs[3] = {r.direction.x(), r.direction.y(), r.direction.z()};
    for (int a = 0; a < 3; ++a) {
      double inv_d = 1.0 / dirs[a];
      double t0 = (mins[a] - orig[a]) * inv_d;
      double t1 = (maxs[a] - orig[a]) * inv_d;
      if (inv_d < 0.0) std::swap(t0, t1);
      t_min = t0 > t_min ? t0 : t_min;
      t_max = t1 < t_max ? t1 : t_max;
      if (t_max <= t_min) return false;
    }
    return true;
  }
  static Aabb surrounding_box(const Aabb& b0, const Aabb& b1) {  // aabb.rs:63-77
    Point3 small(std::fmin(b0.minimum.x(), b1.minimum.x()), std::fmin(b0.minimum.y(), b1.minimum.y()),
                 std::fmin(b0.minimum.z(), b1.minimum.z()));
    Point3 big(std::fmax(b0.maximum.x(), b1.maximum.x()), std::fmax(b0.maximum.y(), b1.maximum.y()),
               std::fmax(b0.maximum.z(), b1.maximum.z()));
    return Aabb(small, big);
  }
};

// ------------------------------------------------------------------ perlin.rs
struct Perlin {
  const rt::FlatPerlin* t;  // tables generated by the builder (Perlin::new, perlin.rs:14-26)
  double noise(const Point3& p) const {  // perlin.rs:28-52
    double u = p.x() - std::floor(p.x());
    double v = p.y() - std::floor(p.y());
    double w = p.z() - std::floor(p.z());
    int32_t i = as_i32(std::floor(p.x()));
    int32_t j = as_i32(std::floor(p.y()));
    int32_t k = as_i32(std::floor(p.z()));
    Vec3 c[2][2][2];
    for (int di = 0; di < 2; ++di)
      for (int dj = 0; dj < 2; ++dj)
        for (int dk = 0; dk < 2; ++dk) {
          int32_t idx = t->perm_x[(uint32_t)((uint32_t)i + (uint32_t)di) & 255u] ^
                        t->perm_y[(uint32_t)((uint32_t)j + (uint32_t)dj) & 255u] ^
                        t->perm_z[(uint32_t)((uint32_t)k + (uint32_t)dk) & 255u];
          c[di][dj][dk] = Vec3(t->ranvec[idx][0], t->ranvec[idx][1], t->ranvec[idx][2]);
        }
    return trilinear_interp(c, u, v, w);
  }
  double turbulence(const Point3& p, int depth) const {  // perlin.rs:54-66
    double accum = 0.0;
    Point3 temp_p = p;
    double weight = 1.0;
    for (int n = 0; n < depth; ++n) {
      accum += weight * noise(temp_p);
      weight *= 0.5;
      temp_p = temp_p * 2.0;
    }
    return std::fabs(accum);
  }
  static double trilinear_interp(const Vec3 c[2][2][2], double u, double v, double w) {  // perlin.rs:85-106
    double uu = u * u * (3.0 - 2.0 * u);
    double vv = v * v * (3.0 - 2.0 * v);
    double ww = w * w * (3.0 - 2.0 * w);
    double accum = 0.0;
    for (int i = 0; i < 2; ++i)
      for (int j = 0; j < 2; ++j)
        for (int k = 0; k < 2; ++k) {
          double i1 = (double)i, j1 = (double)j, k1 = (double)k;
          Vec3 weight_v(u - (double)i, v - (double)j, w - (double)k);
          accum += (i1 * uu + (1.0 - i1) * (1.0 - uu)) * (j1 * vv + (1.0 - j1) * (1.0 - vv)) *
                   (k1 * ww + (1.0 - k1) * (1.0 - ww)) * c[i][j][k].dot(weight_v);
        }
    return accum;
  }
};

// ------------------------------------------------------------------ texture.rs
struct Texture {
  virtual ~Texture() {}
  virtual Color value(double u, double v, const Point3& p) const = 0;
};
struct SolidColor : Texture {  // texture.rs:11-31
  Color c;
  explicit SolidColor(const Color& cc) : c(cc) {}
  Color value(double, double, const Point3&) const override { return c; }
};
struct Checker : Texture {  // texture.rs:33-64
  std::shared_ptr<Texture> even, odd;
  Color value(double u, double v, const Point3& p) const override {
    double sines = rt::rt_sin(10.0 * p.x()) * rt::rt_sin(10.0 * p.y()) * rt::rt_sin(10.0 * p.z());
    if (sines < 0.0) return odd->value(u, v, p);
    return even->value(u, v, p);
  }
};
struct Noise : Texture {  // texture.rs:66-88
  Perlin noise;
  double scale;
  Color value(double, double, const Point3& p) const override {
    return Color(1, 1, 1) * 0.5 * (1.0 + rt::rt_sin(scale * p.z() + 10.0 * noise.turbulence(p, 7)));
  }
};
struct Image : Texture {  // texture.rs:90-122 over a Screen (screen.rs:6-38)
  int32_t width, height;
  const double* pixels;  // Screen.pixels, 3 doubles each
  Color value(double u, double v, const Point3&) const override {
    u = clamp(u, 0.0, 1.0);
    v = 1.0 - clamp(v, 0.0, 1.0);
    int32_t i = as_i32(u * (double)width);
    int32_t j = as_i32(v * (double)height);
    i = std::min(i, width - 1);
    j = std::min(j, height - 1);
    double color_scale = 1.0 / 255.0;
    const double* px = pixels + 3 * ((size_t)j * (size_t)width + (size_t)i);  // Screen::get(j, i), screen.rs:30-33
    return Color(color_scale * px[0], color_scale * px[1], color_scale * px[2]);
  }
};

// ------------------------------------------------------------------ hit.rs: records, traits
struct Material;
struct HitRecord {  // hit.rs:10-18
  Point3 p;
  Vec3 normal;
  double t = 0, u = 0, v = 0;
  bool front_face = false;
  const Material* mat_ptr = nullptr;
};
static void create_normal_face(const Ray& r, const Vec3& outward_normal, Vec3* normal, bool* ff) {  // hit.rs:69-79
  bool front_face = r.direction.dot(outward_normal) < 0.0;
  *normal = front_face ? outward_normal : -outward_normal;
  *ff = front_face;
}
struct Material {  // hit.rs:1013-1018
  virtual ~Material() {}
  virtual bool scatter(const Ray& r_in, const HitRecord& rec, Rng& g, Ray* scattered, Color* attenuation) const = 0;
  virtual Color emitted(double, double, const Point3&) const { return Color(0, 0, 0); }
};
struct Hittable {  // hit.rs:82-85.  The Rng is the path's stream (ConstantMedium draws inside hit).
  virtual ~Hittable() {}
  virtual bool hit(const Ray& r, double t_min, double t_max, HitRecord* rec, Rng& g) const = 0;
  virtual bool bounding_box(double time0, double time1, Aabb* out) const = 0;
};
typedef std::shared_ptr<Hittable> HittablePtr;

// ------------------------------------------------------------------ hit.rs: materials
struct Lambertian : Material {  // hit.rs:1020-1052
  std::shared_ptr<Texture> albedo;
  bool scatter(const Ray& r_in, const HitRecord& rec, Rng& g, Ray* scattered, Color* attenuation) const override {
    Vec3 scatter_direction = rec.normal + random_unit_vector(g);
    if (scatter_direction.near_zero()) scatter_direction = rec.normal;
    *scattered = Ray(rec.p, scatter_direction, r_in.time);
    *attenuation = albedo->value(rec.u, rec.v, rec.p);
    return true;
  }
};
struct Metal : Material {  // hit.rs:1054-1084
  Color albedo;
  double fuzz;
  bool scatter(const Ray& r_in, const HitRecord& rec, Rng& g, Ray* scattered, Color* attenuation) const override {
    Vec3 reflected = reflect(unit(r_in.direction), rec.normal);
    *scattered = Ray(rec.p, reflected + fuzz * random_in_unit_sphere(g), r_in.time);
    if (scattered->direction.dot(rec.normal) > 0.0) { *attenuation = albedo; return true; }
    return false;
  }
};
struct Dielectric : Material {  // hit.rs:1086-1127
  double ir;
  static double reflectance(double cosine, double ref_idx) {  // hit.rs:1095-1099
    double r0 = (1.0 - ref_idx) / (1.0 + ref_idx);
    r0 = r0 * r0;
    double x = 1.0 - cosine;  // powi(x, 5): x * ((x*x) * (x*x)) (LLVM powi / __powidf2 order)
    double x2 = x * x;
    double x4 = x2 * x2;
    return r0 + (1.0 - r0) * (x * x4);
  }
  bool scatter(const Ray& r_in, const HitRecord& rec, Rng& g, Ray* scattered, Color* attenuation) const override {
    *attenuation = Vec3(1, 1, 1);
    double refraction_ratio = rec.front_face ? 1.0 / ir : ir;
    Vec3 unit_direction = unit(r_in.direction);
    double cos_theta = std::fmin((-unit_direction).dot(rec.normal), 1.0);
    double sin_theta = std::sqrt(1.0 - cos_theta * cos_theta);
    bool cannot_refract = refraction_ratio * sin_theta > 1.0;
    Vec3 direction;
    if (cannot_refract || reflectance(cos_theta, refraction_ratio) > rt::rng_f64(g))
      direction = reflect(unit_direction, rec.normal);
    else
      direction = refract(unit_direction, rec.normal, refraction_ratio);
    *scattered = Ray(rec.p, direction, r_in.time);
    return true;
  }
};
struct DiffuseLight : Material {  // hit.rs:1129-1152
  std::shared_ptr<Texture> emit;
  bool scatter(const Ray&, const HitRecord&, Rng&, Ray*, Color*) const override { return false; }
  Color emitted(double u, double v, const Point3& p) const override { return emit->value(u, v, p); }
};
struct Isotropic : Material {  // hit.rs:992-1011
  std::shared_ptr<Texture> albedo;
  bool scatter(const Ray& r_in, const HitRecord& rec, Rng& g, Ray* scattered, Color* attenuation) const override {
    *scattered = Ray(rec.p, random_in_unit_sphere(g), r_in.time);
    *attenuation = albedo->value(rec.u, rec.v, rec.p);
    return true;
  }
};

// ------------------------------------------------------------------ hit.rs: hittables
struct Triangle : Hittable {  // hit.rs:87-178
  Point3 v0, v1, v2, normal;
  const Material* mat;
  bool hit(const Ray& r, double t_min, double t_max, HitRecord* rec, Rng&) const override {
    if (std::fabs(normal.dot(r.direction)) < 0.0001) return false;
    double d = -normal.dot(v0);
    double t = -(normal.dot(r.origin) + d) / normal.dot(r.direction);
    if (t < t_min || t > t_max) return false;
    Point3 p = r.at(t);
    Vec3 edge0 = v1 - v0, vp0 = p - v0;
    Vec3 c = edge0.cross(vp0);
    if (normal.dot(c) < 0.0) return false;
    Vec3 edge1 = v2 - v1, vp1 = p - v1;
    c = edge1.cross(vp1);
    if (normal.dot(c) < 0.0) return false;
    Vec3 edge2 = v0 - v2, vp2 = p - v2;
    c = edge2.cross(vp2);
    if (normal.dot(c) < 0.0) return false;
    create_normal_face(r, normal, &rec->normal, &rec->front_face);
    rec->p = r.at(t); rec->t = t; rec->u = 1.0; rec->v = 1.0; rec->mat_ptr = mat;
    return true;
  }
  bool bounding_box(double, double, Aabb* out) const override {  // hit.rs:164-177
    double inf = std::numeric_limits<double>::infinity();
    double mn[3] = {inf, inf, inf}, mx[3] = {-inf, -inf, -inf};
    const Point3* vs[3] = {&v0, &v1, &v2};
    for (const Point3* v : vs) {
      mn[0] = std::fmin(mn[0], v->x()); mn[1] = std::fmin(mn[1], v->y()); mn[2] = std::fmin(mn[2], v->z());
      mx[0] = std::fmax(mx[0], v->x()); mx[1] = std::fmax(mx[1], v->y()); mx[2] = std::fmax(mx[2], v->z());
    }
    *out = Aabb(Point3(mn[0], mn[1], mn[2]), Point3(mx[0], mx[1], mx[2]));
    return true;
  }
};
static void get_sphere_uv(const Point3& p, double* u, double* v) {  // hit.rs:195-200
  double theta = rt::rt_acos(-p.y());
  double phi = rt::rt_atan2(-p.z(), p.x()) + RT_PI;
  *u = phi / (2.0 * RT_PI);
  *v = theta / RT_PI;
}
struct Sphere : Hittable {  // hit.rs:180-245
  Point3 center;
  double radius;
  const Material* mat;
  bool hit(const Ray& r, double t_min, double t_max, HitRecord* rec, Rng&) const override {
    Vec3 oc = r.origin - center;
    double a = r.direction.length_squared();
    double half_b = oc.dot(r.direction);
    double c = oc.length_squared() - radius * radius;
    double discriminant = half_b * half_b - a * c;
    if (discriminant < 0.0) return false;
    double sqrtd = std::sqrt(discriminant);
    double root = (-half_b - sqrtd) / a;
    if (root < t_min || t_max < root) {
      root = (-half_b + sqrtd) / a;
      if (root < t_min || t_max < root) return false;
    }
    double t = root;
    Point3 p = r.at(t);
    Vec3 outward_normal = (p - center) / radius;
    create_normal_face(r, outward_normal, &rec->normal, &rec->front_face);
    get_sphere_uv(outward_normal, &rec->u, &rec->v);
    rec->p = p; rec->t = t; rec->mat_ptr = mat;
    return true;
  }
  bool bounding_box(double, double, Aabb* out) const override {  // hit.rs:239-244
    *out = Aabb(center - Point3(radius, radius, radius), center + Point3(radius, radius, radius));
    return true;
  }
};
struct MovingSphere : Hittable {  // hit.rs:247-328
  Point3 center0, center1;
  double time0, time1, radius;
  const Material* mat;
  Point3 get_center(double time) const {  // hit.rs:275-278
    return center0 + ((time - time0) / (time1 - time0)) * (center1 - center0);
  }
  bool hit(const Ray& r, double t_min, double t_max, HitRecord* rec, Rng&) const override {
    Point3 cur = get_center(r.time);
    Vec3 oc = r.origin - cur;
    double a = r.direction.length_squared();
    double half_b = oc.dot(r.direction);
    double c = oc.length_squared() - radius * radius;
    double discriminant = half_b * half_b - a * c;
    if (discriminant < 0.0) return false;
    double sqrtd = std::sqrt(discriminant);
    double root = (-half_b - sqrtd) / a;
    if (root < t_min || t_max < root) {
      root = (-half_b + sqrtd) / a;
      if (root < t_min || t_max < root) return false;
    }
    double t = root;
    Point3 p = r.at(t);
    Vec3 outward_normal = (p - cur) / radius;
    create_normal_face(r, outward_normal, &rec->normal, &rec->front_face);
    rec->p = p; rec->t = t; rec->u = 0.0; rec->v = 0.0; rec->mat_ptr = mat;
    return true;
  }
  bool bounding_box(double t0, double t1, Aabb* out) const override {  // hit.rs:317-327
    Point3 r3(radius, radius, radius);
    Aabb box0(get_center(t0) - r3, get_center(t0) + r3);
    Aabb box1(get_center(t1) - r3, get_center(t1) + r3);
    *out = Aabb::surrounding_box(box0, box1);
    return true;
  }
};
struct GravitySphere : Hittable {  // hit.rs:330-444
  Point3 start;
  double time0, radius;
  const Material* mat;
  std::vector<double> stored;
  // GravitySphere::new, hit.rs:340-367 (the oracle simulates its own table; the product's comes from scene_graph.cpp)
  void simulate() {
    stored.clear();
    stored.push_back(start.y());
    const double incr = 0.001;
    double t = time0, y = start.y(), vel = 0.0;
    while (t < 100.0) {
      t += incr;
      vel -= 0.000001;
      if (y - 1.0 * radius <= 0.0) vel *= -0.92;
      y = std::fmax(1.0 * radius, y + vel);
      stored.push_back(y);
    }
  }
  static size_t as_usize(double x) {  // Rust `as usize`: saturating, NaN -> 0
    if (!(x == x) || x <= 0.0) return 0;
    if (x >= 18446744073709551615.0) return ~(size_t)0;
    return (size_t)x;
  }
  Point3 get_center(double time) const {  // hit.rs:369-391
    const double incr = 0.001;
    const size_t idx = as_usize(time / incr);
    if (idx != ~(size_t)0 && idx + 1 <= stored.size()) return Point3(start.x(), stored[idx], start.z());
    double t = time0, y = start.y(), vel = 0.0;
    while (t < time) {
      t += incr;
      vel -= 0.000001;
      if (y - 2.0 * radius <= 0.0) vel *= -0.8;
      y = std::fmax(2.0 * radius, y + vel);
    }
    return Point3(start.x(), y, start.z());
  }
  bool hit(const Ray& r, double t_min, double t_max, HitRecord* rec, Rng&) const override {  // hit.rs:394-428
    Point3 cur = get_center(r.time);
    Vec3 oc = r.origin - cur;
    double a = r.direction.length_squared();
    double half_b = oc.dot(r.direction);
    double c = oc.length_squared() - radius * radius;
    double discriminant = half_b * half_b - a * c;
    if (discriminant < 0.0) return false;
    double sqrtd = std::sqrt(discriminant);
    double root = (-half_b - sqrtd) / a;
    if (root < t_min || t_max < root) {
      root = (-half_b + sqrtd) / a;
      if (root < t_min || t_max < root) return false;
    }
    double t = root;
    Point3 p = r.at(t);
    Vec3 outward_normal = (p - cur) / radius;
    create_normal_face(r, outward_normal, &rec->normal, &rec->front_face);
    rec->p = p; rec->t = t; rec->u = 0.0; rec->v = 0.0; rec->mat_ptr = mat;
    return true;
  }
  bool bounding_box(double t0, double t1, Aabb* out) const override {  // hit.rs:430-443
    Point3 r3(radius, radius, radius);
    Aabb box0(get_center(t0) - r3, get_center(t0) + r3);
    Aabb box1(get_center(t1) - r3, get_center(t1) + r3);
    *out = Aabb::surrounding_box(box0, box1);
    return true;
  }
};
struct XyRect : Hittable {  // hit.rs:446-509
  double x0, x1, y0, y1, k;
  const Material* mat;
  bool hit(const Ray& r, double t_min, double t_max, HitRecord* rec, Rng&) const override {
    double t = (k - r.origin.z()) / r.direction.z();
    if (t < t_min || t > t_max) return false;
    double x = r.origin.x() + t * r.direction.x();
    double y = r.origin.y() + t * r.direction.y();
    if (x < x0 || x > x1 || y < y0 || y > y1) return false;
    rec->u = (x - x0) / (x1 - x0);
    rec->v = (y - y0) / (y1 - y0);
    create_normal_face(r, Vec3(0, 0, 1), &rec->normal, &rec->front_face);
    rec->p = r.at(t); rec->t = t; rec->mat_ptr = mat;
    return true;
  }
  bool bounding_box(double, double, Aabb* out) const override {
    *out = Aabb(Point3(x0, y0, k - 0.0001), Point3(x1, y1, k + 0.0001));
    return true;
  }
};
struct XzRect : Hittable {  // hit.rs:511-574
  double x0, x1, y0, y1, k;
  const Material* mat;
  bool hit(const Ray& r, double t_min, double t_max, HitRecord* rec, Rng&) const override {
    double t = (k - r.origin.y()) / r.direction.y();
    if (t < t_min || t > t_max) return false;
    double x = r.origin.x() + t * r.direction.x();
    double y = r.origin.z() + t * r.direction.z();
    if (x < x0 || x > x1 || y < y0 || y > y1) return false;
    rec->u = (x - x0) / (x1 - x0);
    rec->v = (y - y0) / (y1 - y0);
    create_normal_face(r, Vec3(0, 1, 0), &rec->normal, &rec->front_face);
    rec->p = r.at(t); rec->t = t; rec->mat_ptr = mat;
    return true;
  }
  bool bounding_box(double, double, Aabb* out) const override {
    *out = Aabb(Point3(x0, k - 0.0001, y0), Point3(x1, k + 0.0001, y1));
    return true;
  }
};
struct YzRect : Hittable {  // hit.rs:576-639
  double x0, x1, y0, y1, k;
  const Material* mat;
  bool hit(const Ray& r, double t_min, double t_max, HitRecord* rec, Rng&) const override {
    double t = (k - r.origin.x()) / r.direction.x();
    if (t < t_min || t > t_max) return false;
    double x = r.origin.y() + t * r.direction.y();
    double y = r.origin.z() + t * r.direction.z();
    if (x < x0 || x > x1 || y < y0 || y > y1) return false;
    rec->u = (x - x0) / (x1 - x0);
    rec->v = (y - y0) / (y1 - y0);
    create_normal_face(r, Vec3(1, 0, 0), &rec->normal, &rec->front_face);
    rec->p = r.at(t); rec->t = t; rec->mat_ptr = mat;
    return true;
  }
  bool bounding_box(double, double, Aabb* out) const override {
    *out = Aabb(Point3(k - 0.0001, x0, y0), Point3(k + 0.0001, x1, y1));
    return true;
  }
};
struct HittableList : Hittable {  // hit.rs:641-711
  std::vector<HittablePtr> objects;
  bool hit(const Ray& r, double t_min, double t_max, HitRecord* rec, Rng& g) const override {
    bool hit_anything = false;
    double closest_so_far = t_max;
    HitRecord temp_rec;
    for (const HittablePtr& object : objects) {
      HitRecord one;
      if (object->hit(r, t_min, closest_so_far, &one, g)) {
        hit_anything = true;
        closest_so_far = one.t;
        temp_rec = one;
      }
    }
    if (hit_anything) *rec = temp_rec;
    return hit_anything;
  }
  bool bounding_box(double t0, double t1, Aabb* out) const override {
    if (objects.empty()) return false;
    Aabb temp_box;
    if (!objects[0]->bounding_box(t0, t1, &temp_box)) return false;
    for (size_t i = 1; i < objects.size(); ++i) {
      Aabb other;
      if (!objects[i]->bounding_box(t0, t1, &other)) return false;
      temp_box = Aabb::surrounding_box(temp_box, other);
    }
    *out = temp_box;
    return true;
  }
};
struct RectPrism : Hittable {  // hit.rs:713-785
  Point3 box_min, box_max;
  HittableList sides;
  bool hit(const Ray& r, double t_min, double t_max, HitRecord* rec, Rng& g) const override {
    return sides.hit(r, t_min, t_max, rec, g);
  }
  bool bounding_box(double, double, Aabb* out) const override { *out = Aabb(box_min, box_max); return true; }
};
struct Translate : Hittable {  // hit.rs:787-833
  HittablePtr obj;
  Vec3 offset;
  bool hit(const Ray& r, double t_min, double t_max, HitRecord* rec, Rng& g) const override {
    Ray moved_r(r.origin - offset, r.direction, r.time);
    HitRecord in;
    if (!obj->hit(moved_r, t_min, t_max, &in, g)) return false;
    create_normal_face(moved_r, in.normal, &rec->normal, &rec->front_face);
    rec->p = in.p + offset; rec->t = in.t; rec->u = in.u; rec->v = in.v; rec->mat_ptr = in.mat_ptr;
    return true;
  }
  bool bounding_box(double t0, double t1, Aabb* out) const override {
    Aabb a;
    if (!obj->bounding_box(t0, t1, &a)) return false;
    *out = Aabb(a.minimum + offset, a.maximum + offset);
    return true;
  }
};
struct RotateY : Hittable {  // hit.rs:835-936
  HittablePtr obj;
  double sin_theta, cos_theta;
  bool has_box;
  Aabb bbox;  // hit.rs:886: the UN-rotated child box is what gets stored
  bool hit(const Ray& r, double t_min, double t_max, HitRecord* rec, Rng& g) const override {
    Vec3 origin(cos_theta * r.origin.x() - sin_theta * r.origin.z(), r.origin.y(),
                sin_theta * r.origin.x() + cos_theta * r.origin.z());
    Vec3 direction(cos_theta * r.direction.x() - sin_theta * r.direction.z(), r.direction.y(),
                   sin_theta * r.direction.x() + cos_theta * r.direction.z());
    Ray rotated_r(origin, direction, r.time);
    HitRecord in;
    if (!obj->hit(rotated_r, t_min, t_max, &in, g)) return false;
    Vec3 p(cos_theta * in.p.x() + sin_theta * in.p.z(), in.p.y(), -sin_theta * in.p.x() + cos_theta * in.p.z());
    Vec3 normal(cos_theta * in.normal.x() + sin_theta * in.normal.z(), in.normal.y(),
                -sin_theta * in.normal.x() + cos_theta * in.normal.z());
    create_normal_face(rotated_r, normal, &rec->normal, &rec->front_face);
    rec->p = p; rec->t = in.t; rec->u = in.u; rec->v = in.v; rec->mat_ptr = in.mat_ptr;
    return true;
  }
  bool bounding_box(double, double, Aabb* out) const override {
    if (!has_box) return false;
    *out = bbox;
    return true;
  }
};
struct ConstantMedium : Hittable {  // hit.rs:938-990
  HittablePtr boundary;
  const Material* phase_function;
  double neg_inv_density;
  bool hit(const Ray& r, double t_min, double t_max, HitRecord* rec, Rng& g) const override {
    const double inf = std::numeric_limits<double>::infinity();
    HitRecord rec1, rec2;
    if (!boundary->hit(r, -inf, inf, &rec1, g)) return false;
    if (!boundary->hit(r, rec1.t + 0.0001, inf, &rec2, g)) return false;
    double t1 = std::fmax(rec1.t, t_min);
    double t2 = std::fmin(rec2.t, t_max);
    if (t1 >= t2) return false;
    if (t1 < 0.0) t1 = 0.0;
    double ray_length = r.direction.length();
    double distance_inside_boundary = (t2 - t1) * ray_length;
    double hit_distance = neg_inv_density * rt::rt_log(rt::rng_f64(g));
    if (hit_distance > distance_inside_boundary) return false;
    double t = t1 + hit_distance / ray_length;
    rec->p = r.at(t); rec->normal = Vec3(0, 0, 0); rec->t = t; rec->u = 0.0; rec->v = 0.0;
    rec->front_face = true; rec->mat_ptr = phase_function;
    return true;
  }
  bool bounding_box(double t0, double t1, Aabb* out) const override { return boundary->bounding_box(t0, t1, out); }
};

// ------------------------------------------------------------------ bvh.rs
struct BvhNode : Hittable {
  HittablePtr left, right;
  Aabb bbox;
  // bvh.rs:14-83.  The reference clones the whole Vec at every node; cloning only the
  // [start, end) slice it then sorts gives the same tree in O(n log^2 n).  Axis is drawn from
  // a seeded stream in place of thread_rng().gen_range(0..2) (z is never chosen, bvh.rs:24).
  static std::shared_ptr<BvhNode> build(std::vector<HittablePtr>& objects, size_t start, size_t end,
                                        double time0, double time1, rt::HostRng& rng) {
    auto node = std::make_shared<BvhNode>();
    int axis = (int)rt::host_rng_below(rng, 2);
    auto box_less = [axis](const HittablePtr& a, const HittablePtr& b) {  // bvh.rs:25-46
      Aabb box_a, box_b;
      a->bounding_box(0.0, 0.0, &box_a);
      b->bounding_box(0.0, 0.0, &box_b);
      if (axis == 0) return box_a.minimum.x() < box_b.minimum.x();
      return box_a.minimum.y() < box_b.minimum.y();
    };
    size_t object_span = end - start;
    if (object_span == 1) {
      node->left = objects[start];
      node->right = objects[start];
    } else if (object_span == 2) {
      if (box_less(objects[start], objects[start + 1])) { node->left = objects[start]; node->right = objects[start + 1]; }
      else { node->left = objects[start + 1]; node->right = objects[start]; }
    } else {
      std::vector<HittablePtr> local(objects.begin() + (long)start, objects.begin() + (long)end);
      std::stable_sort(local.begin(), local.end(), box_less);  // Rust sort_by is a stable merge sort
      size_t mid = object_span / 2;
      node->left = build(local, 0, mid, time0, time1, rng);
      node->right = build(local, mid, object_span, time0, time1, rng);
    }
    Aabb lb, rb;
    node->left->bounding_box(time0, time1, &lb);
    node->right->bounding_box(time0, time1, &rb);
    node->bbox = Aabb::surrounding_box(lb, rb);
    return node;
  }
  bool hit(const Ray& r, double t_min, double t_max, HitRecord* rec, Rng& g) const override {  // bvh.rs:97-112
    if (!bbox.hit(r, t_min, t_max)) return false;
    HitRecord l;
    if (left->hit(r, t_min, t_max, &l, g)) {
      HitRecord rr;
      if (right->hit(r, t_min, l.t, &rr, g)) { *rec = rr; return true; }
      *rec = l;
      return true;
    }
    return right->hit(r, t_min, t_max, rec, g);
  }
  bool bounding_box(double, double, Aabb* out) const override { *out = bbox; return true; }
};

// ------------------------------------------------------------------ camera.rs
struct Camera {  // camera.rs:6-17; fields arrive already derived by Camera::new (host)
  Point3 origin, lower_left_corner;
  Vec3 horizontal, vertical, u, v, w;
  double lens_radius, time1, time2;
  Ray get_ray(double s, double t, Rng& g) const {  // camera.rs:59-71
    Vec3 rd = lens_radius * random_in_unit_disk(g);
    Vec3 offset = u * rd.x() + v * rd.y();
    double time = 0;
    Point3 o = origin + offset;
    Vec3 d = lower_left_corner + s * horizontal + t * vertical - origin - offset;
    time = rt::rng_range(g, time1, time2);
    return Ray(o, d, time);
  }
};

// ------------------------------------------------------------------ world.rs:52-93
static Color ray_color(const Ray& r, const Color& background, const Hittable& world, int depth, Rng& g) {
  Vec3 product(1, 1, 1);
  Vec3 output(0, 0, 0);
  Ray current_ray = r;
  for (;;) {
    depth -= 1;
    if (depth < 0) break;
    HitRecord rec;
    if (world.hit(current_ray, 0.001, std::numeric_limits<double>::infinity(), &rec, g)) {
      Ray scattered;
      Color attenuation;
      if (rec.mat_ptr->scatter(current_ray, rec, g, &scattered, &attenuation)) {
        Color emitted = rec.mat_ptr->emitted(rec.u, rec.v, rec.p);
        output += emitted * product;
        product *= attenuation;
        current_ray = scattered;
      } else {
        Color emitted = rec.mat_ptr->emitted(rec.u, rec.v, rec.p);
        output += emitted * product;
        break;
      }
    } else {
      output += product * background;
      break;
    }
  }
  return output;
}

// ------------------------------------------------------------------ graph -> object graph
struct World {
  std::vector<std::shared_ptr<Texture>> textures;
  std::vector<std::shared_ptr<Material>> materials;
  std::unordered_map<int32_t, HittablePtr> cache;
  const rtx::SceneGraph* g = nullptr;
  rt::HostRng bvh_rng;
  std::string err;

  std::shared_ptr<Texture> texture(int32_t h) {
    if (textures[h]) return textures[h];
    const rtx::GTexture& t = g->textures[h];
    std::shared_ptr<Texture> out;
    if (t.kind == rt::TEX_SOLID) out = std::make_shared<SolidColor>(Color(t.color[0], t.color[1], t.color[2]));
    else if (t.kind == rt::TEX_CHECKER) { auto c = std::make_shared<Checker>(); c->even = texture(t.a); c->odd = texture(t.b); out = c; }
    else if (t.kind == rt::TEX_NOISE) { auto n = std::make_shared<Noise>(); n->noise.t = &g->perlins[t.a]; n->scale = t.scale; out = n; }
    else { auto im = std::make_shared<Image>(); const rtx::GImage& gi = g->images[t.a]; im->width = gi.width; im->height = gi.height; im->pixels = gi.texels.data(); out = im; }
    textures[h] = out;
    return out;
  }
  const Material* material(int32_t h) {
    if (materials[h]) return materials[h].get();
    const rtx::GMaterial& m = g->materials[h];
    std::shared_ptr<Material> out;
    if (m.kind == rt::MAT_LAMBERTIAN) { auto x = std::make_shared<Lambertian>(); x->albedo = texture(m.tex); out = x; }
    else if (m.kind == rt::MAT_METAL) { auto x = std::make_shared<Metal>(); x->albedo = Color(m.albedo[0], m.albedo[1], m.albedo[2]); x->fuzz = m.param; out = x; }
    else if (m.kind == rt::MAT_DIELECTRIC) { auto x = std::make_shared<Dielectric>(); x->ir = m.param; out = x; }
    else if (m.kind == rt::MAT_DIFFUSE_LIGHT) { auto x = std::make_shared<DiffuseLight>(); x->emit = texture(m.tex); out = x; }
    else { auto x = std::make_shared<Isotropic>(); x->albedo = texture(m.tex); out = x; }
    materials[h] = out;
    return out.get();
  }
  template <class R>
  std::shared_ptr<R> rect(const rtx::GHittable& o) {
    auto q = std::make_shared<R>();
    q->x0 = o.f[0]; q->x1 = o.f[1]; q->y0 = o.f[2]; q->y1 = o.f[3]; q->k = o.f[4]; q->mat = material(o.mat);
    return q;
  }
  template <class R>
  static std::shared_ptr<R> side(double a0, double a1, double b0, double b1, double k, const Material* m) {
    auto q = std::make_shared<R>();
    q->x0 = a0; q->x1 = a1; q->y0 = b0; q->y1 = b1; q->k = k; q->mat = m;
    return q;
  }
  HittablePtr hittable(int32_t h) {
    auto it = cache.find(h);
    if (it != cache.end()) return it->second;
    const rtx::GHittable& o = g->hittables[h];
    HittablePtr out;
    switch (o.kind) {
      case rtx::H_SPHERE: { auto s = std::make_shared<Sphere>(); s->center = Point3(o.f[0], o.f[1], o.f[2]); s->radius = o.f[3]; s->mat = material(o.mat); out = s; break; }
      case rtx::H_MOVING_SPHERE: {
        auto s = std::make_shared<MovingSphere>();
        s->center0 = Point3(o.f[0], o.f[1], o.f[2]); s->center1 = Point3(o.f[3], o.f[4], o.f[5]);
        s->time0 = o.f[6]; s->time1 = o.f[7]; s->radius = o.f[8]; s->mat = material(o.mat);
        out = s; break;
      }
      case rtx::H_GRAVITY_SPHERE: {
        auto s = std::make_shared<GravitySphere>();
        s->start = Point3(o.f[0], o.f[1], o.f[2]); s->time0 = o.f[3]; s->radius = o.f[4]; s->mat = material(o.mat);
        s->simulate();
        out = s; break;
      }
      case rtx::H_TRIANGLE: {  // Triangle::new, hit.rs:96-107
        auto t = std::make_shared<Triangle>();
        t->v0 = Point3(o.f[0], o.f[1], o.f[2]); t->v1 = Point3(o.f[3], o.f[4], o.f[5]); t->v2 = Point3(o.f[6], o.f[7], o.f[8]);
        Vec3 a = t->v1 - t->v0, b = t->v2 - t->v0;
        t->normal = unit(a.cross(b));
        t->mat = material(o.mat);
        out = t; break;
      }
      case rtx::H_XY_RECT: out = rect<XyRect>(o); break;
      case rtx::H_XZ_RECT: out = rect<XzRect>(o); break;
      case rtx::H_YZ_RECT: out = rect<YzRect>(o); break;
      case rtx::H_LIST: {
        auto l = std::make_shared<HittableList>();
        for (int32_t c : o.children) l->objects.push_back(hittable(c));
        out = l; break;
      }
      case rtx::H_RECT_PRISM: {  // RectPrism::new, hit.rs:720-775
        auto p = std::make_shared<RectPrism>();
        Point3 p0(o.f[0], o.f[1], o.f[2]), p1(o.f[3], o.f[4], o.f[5]);
        const Material* m = material(o.mat);
        p->box_min = p0; p->box_max = p1;
        p->sides.objects.push_back(side<XyRect>(p0.x(), p1.x(), p0.y(), p1.y(), p1.z(), m));
        p->sides.objects.push_back(side<XyRect>(p0.x(), p1.x(), p0.y(), p1.y(), p0.z(), m));
        p->sides.objects.push_back(side<XzRect>(p0.x(), p1.x(), p0.z(), p1.z(), p1.y(), m));
        p->sides.objects.push_back(side<XzRect>(p0.x(), p1.x(), p0.z(), p1.z(), p0.y(), m));
        p->sides.objects.push_back(side<YzRect>(p0.y(), p1.y(), p0.z(), p1.z(), p1.x(), m));
        p->sides.objects.push_back(side<YzRect>(p0.y(), p1.y(), p0.z(), p1.z(), p0.x(), m));
        out = p; break;
      }
      case rtx::H_BVH: {
        std::vector<HittablePtr> objs;
        for (int32_t c : o.children) objs.push_back(hittable(c));
        out = BvhNode::build(objs, 0, objs.size(), o.f[0], o.f[1], bvh_rng);
        break;
      }
      case rtx::H_TRANSLATE: {
        auto t = std::make_shared<Translate>();
        t->obj = hittable(o.children[0]); t->offset = Vec3(o.f[0], o.f[1], o.f[2]);
        out = t; break;
      }
      case rtx::H_ROTATE_Y: {  // RotateY::new, hit.rs:843-888 (sin/cos computed by the builder with rt_math)
        auto r = std::make_shared<RotateY>();
        r->obj = hittable(o.children[0]); r->sin_theta = o.f[0]; r->cos_theta = o.f[1];
        r->has_box = r->obj->bounding_box(0.0, 1.0, &r->bbox);
        out = r; break;
      }
      default: {  // H_CONSTANT_MEDIUM, hit.rs:945-951
        auto m = std::make_shared<ConstantMedium>();
        m->boundary = hittable(o.children[0]); m->phase_function = material(o.mat); m->neg_inv_density = o.f[0];
        out = m; break;
      }
    }
    cache[h] = out;
    return out;
  }
};

static Camera to_camera(const OracleCamera* c) {
  Camera k;
  k.origin = Point3(c->origin[0], c->origin[1], c->origin[2]);
  k.lower_left_corner = Point3(c->lower_left_corner[0], c->lower_left_corner[1], c->lower_left_corner[2]);
  k.horizontal = Vec3(c->horizontal[0], c->horizontal[1], c->horizontal[2]);
  k.vertical = Vec3(c->vertical[0], c->vertical[1], c->vertical[2]);
  k.u = Vec3(c->u[0], c->u[1], c->u[2]);
  k.v = Vec3(c->v[0], c->v[1], c->v[2]);
  k.w = Vec3(c->w[0], c->w[1], c->w[2]);
  k.lens_radius = c->lens_radius; k.time1 = c->time1; k.time2 = c->time2;
  return k;
}

}  // namespace o1

using namespace o1;

extern "C" {

// render_scene (world.rs:1181-1247) with the counter RNG: row bands over `threads` OS threads,
// per pixel a sequential sum over samples, then get_normalized_color.
int oracle_o1_render(const void* graph_ptr, int32_t world_handle, const OracleCamera* cam,
                     const OracleConfig* cfg, double* accum_rgb, uint8_t* rgb8) {
  if (!graph_ptr || !cam || !cfg) return 1;
  const rtx::SceneGraph* g = (const rtx::SceneGraph*)graph_ptr;
  if (!g->valid_hittable(world_handle)) return 1;
  World wb;
  wb.g = g;
  wb.textures.resize(g->textures.size());
  wb.materials.resize(g->materials.size());
  wb.bvh_rng.state = cfg->bvh_seed;
  HittablePtr world = wb.hittable(world_handle);
  Camera camera = to_camera(cam);
  const int32_t w = cfg->image_width, h = cfg->image_height;
  const int32_t spp = cfg->samples_per_pixel, max_depth = cfg->max_depth;
  const Color background(cfg->background[0], cfg->background[1], cfg->background[2]);
  int threads = cfg->threads > 0 ? cfg->threads : 1;
  if (accum_rgb) memset(accum_rgb, 0, sizeof(double) * 3 * (size_t)w * h);  // Screen::new: all (0,0,0)
  if (rgb8) memset(rgb8, 0, 3 * (size_t)w * h);
  // world.rs:1198-1202: chunk_size = h / threads; band t = [t*chunk, t*chunk + chunk).
  // With row_chunk_compat == 0 the remainder rows are appended to the last band (this build
  // renders every row; the reference silently leaves them black).
  size_t chunk_size = (size_t)h / (size_t)threads;
  std::vector<std::thread> pool;
  for (int t = 0; t < threads; ++t) {
    size_t start = (size_t)t * chunk_size;
    size_t end = std::min((size_t)t * chunk_size + chunk_size, (size_t)h);
    if (!cfg->row_chunk_compat && t == threads - 1) end = (size_t)h;
    pool.emplace_back([=, &camera, &world]() {
      for (size_t j = start; j < end; ++j) {
        for (int32_t i = 0; i < w; ++i) {
          Vec3 pixel(0, 0, 0);
          uint64_t pixel_index = (uint64_t)j * (uint64_t)w + (uint64_t)i;
          for (int32_t s = 0; s < spp; ++s) {
            Rng rng = rt::rng_for_sample(cfg->seed, pixel_index, (uint32_t)s);
            double ru = rt::rng_f64(rng);
            double u = ((double)i + ru) / (double)(w - 1);  // world.rs:1212
            double rv = rt::rng_f64(rng);
            double v = ((double)j + rv) / (double)(h - 1);  // world.rs:1213
            Ray r = camera.get_ray(u, v, rng);
            pixel += ray_color(r, background, *world, max_depth, rng);
          }
          size_t o = 3 * ((size_t)j * (size_t)w + (size_t)i);
          if (accum_rgb) { accum_rgb[o] = pixel.x(); accum_rgb[o + 1] = pixel.y(); accum_rgb[o + 2] = pixel.z(); }
          if (rgb8) {
            int32_t c[3];
            get_normalized_color(pixel, (uint32_t)spp, c);
            rgb8[o] = (uint8_t)c[0]; rgb8[o + 1] = (uint8_t)c[1]; rgb8[o + 2] = (uint8_t)c[2];
          }
        }
      }
    });
  }
  for (std::thread& th : pool) th.join();
  return 0;
}

// The same per-pixel loop over a rectangle of the frame only: rows [j0, j1) counted from the BOTTOM row, columns [i0, i1).
// A pixel is a pure function of (scene, camera, config, i, j), so the window equals that part of the whole frame; it lets a test
// spend its samples on the regions the reference's images pin (tests/test_reference_image_pins.py) instead of on 10^6 pixels.
// accum_rgb: (j1 - j0) x (i1 - i0) x 3, the window's bottom row first.  Pixels are dealt to the threads one by one.
int oracle_o1_render_window(const void* graph_ptr, int32_t world_handle, const OracleCamera* cam, const OracleConfig* cfg,
                            int32_t j0, int32_t j1, int32_t i0, int32_t i1, double* accum_rgb) {
  if (!graph_ptr || !cam || !cfg || !accum_rgb) return 1;
  const rtx::SceneGraph* g = (const rtx::SceneGraph*)graph_ptr;
  if (!g->valid_hittable(world_handle)) return 1;
  const int32_t w = cfg->image_width, h = cfg->image_height;
  if (j0 < 0 || j1 > h || i0 < 0 || i1 > w || j0 >= j1 || i0 >= i1) return 1;
  World wb;
  wb.g = g;
  wb.textures.resize(g->textures.size());
  wb.materials.resize(g->materials.size());
  wb.bvh_rng.state = cfg->bvh_seed;
  HittablePtr world = wb.hittable(world_handle);
  Camera camera = to_camera(cam);
  const int32_t spp = cfg->samples_per_pixel, max_depth = cfg->max_depth;
  const Color background(cfg->background[0], cfg->background[1], cfg->background[2]);
  const int threads = cfg->threads > 0 ? cfg->threads : 1;
  const int64_t ww = i1 - i0, n_pix = (int64_t)(j1 - j0) * ww;
  std::atomic<int64_t> next{0};
  std::vector<std::thread> pool;
  for (int t = 0; t < threads; ++t) {
    pool.emplace_back([&]() {
      for (;;) {
        const int64_t p = next.fetch_add(1);
        if (p >= n_pix) break;
        const int32_t j = j0 + (int32_t)(p / ww), i = i0 + (int32_t)(p % ww);
        Vec3 pixel(0, 0, 0);
        const uint64_t pixel_index = (uint64_t)j * (uint64_t)w + (uint64_t)i;
        for (int32_t s = 0; s < spp; ++s) {
          Rng rng = rt::rng_for_sample(cfg->seed, pixel_index, (uint32_t)s);
          double ru = rt::rng_f64(rng);
          double u = ((double)i + ru) / (double)(w - 1);  // world.rs:1212
          double rv = rt::rng_f64(rng);
          double v = ((double)j + rv) / (double)(h - 1);  // world.rs:1213
          Ray r = camera.get_ray(u, v, rng);
          pixel += ray_color(r, background, *world, max_depth, rng);
        }
        accum_rgb[3 * p] = pixel.x(); accum_rgb[3 * p + 1] = pixel.y(); accum_rgb[3 * p + 2] = pixel.z();
      }
    });
  }
  for (std::thread& th : pool) th.join();
  return 0;
}

// ---- probes for the known-answer tests -------------------------------
void oracle_o1_vec3_ops(const double a[3], const double b[3], double t, double out[24]) {
  Vec3 A(a[0], a[1], a[2]), B(b[0], b[1], b[2]);
  Vec3 r;
  int k = 0;
  auto put = [&](const Vec3& v) { out[k++] = v.x(); out[k++] = v.y(); out[k++] = v.z(); };
  put(A + B); put(A - B); put(A * B); put(A * t); put(A / t); put(-A); put(A.cross(B));
  out[k++] = A.dot(B); out[k++] = A.length_squared(); out[k++] = A.length();
}
void oracle_o1_tone_map(const double sum[3], uint32_t spp, int32_t out[3]) {
  get_normalized_color(Color(sum[0], sum[1], sum[2]), spp, out);
}
void oracle_o1_sphere_uv(const double p[3], double uv[2]) { get_sphere_uv(Point3(p[0], p[1], p[2]), &uv[0], &uv[1]); }
double oracle_o1_reflectance(double cosine, double ref_idx) { return Dielectric::reflectance(cosine, ref_idx); }
void oracle_o1_refract(const double uv[3], const double n[3], double ratio, double out[3]) {
  Vec3 r = refract(Vec3(uv[0], uv[1], uv[2]), Vec3(n[0], n[1], n[2]), ratio);
  out[0] = r.x(); out[1] = r.y(); out[2] = r.z();
}
void oracle_o1_reflect(const double v[3], const double n[3], double out[3]) {
  Vec3 r = reflect(Vec3(v[0], v[1], v[2]), Vec3(n[0], n[1], n[2]));
  out[0] = r.x(); out[1] = r.y(); out[2] = r.z();
}
int oracle_o1_aabb_hit(const double mn[3], const double mx[3], const double o[3], const double d[3], double t_min, double t_max) {
  return Aabb(Point3(mn[0], mn[1], mn[2]), Point3(mx[0], mx[1], mx[2])).hit(Ray(Point3(o[0], o[1], o[2]), Vec3(d[0], d[1], d[2]), 0.0), t_min, t_max) ? 1 : 0;
}
// Hit one object of a graph with one ray: out = {t, p.xyz, n.xyz, u, v, front_face}
int oracle_o1_hit(const void* graph_ptr, int32_t handle, const double o[3], const double d[3], double time,
                  double t_min, double t_max, uint64_t rng_seed, double out[10]) {
  const rtx::SceneGraph* g = (const rtx::SceneGraph*)graph_ptr;
  if (!g || !g->valid_hittable(handle)) return -1;
  World wb;
  wb.g = g;
  wb.textures.resize(g->textures.size());
  wb.materials.resize(g->materials.size());
  wb.bvh_rng.state = 7;
  HittablePtr obj = wb.hittable(handle);
  Rng rng = rt::rng_for_sample(rng_seed, 0, 0);
  HitRecord rec;
  if (!obj->hit(Ray(Point3(o[0], o[1], o[2]), Vec3(d[0], d[1], d[2]), time), t_min, t_max, &rec, rng)) return 0;
  out[0] = rec.t; out[1] = rec.p.x(); out[2] = rec.p.y(); out[3] = rec.p.z();
  out[4] = rec.normal.x(); out[5] = rec.normal.y(); out[6] = rec.normal.z();
  out[7] = rec.u; out[8] = rec.v; out[9] = rec.front_face ? 1.0 : 0.0;
  return 1;
}

}  // extern "C"
