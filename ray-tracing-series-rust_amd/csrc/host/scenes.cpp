// Camera::new and the scene catalogue.  Each builder follows the reference builder it
// names line by line (object order, constructor arguments, order of random draws); the
// random draws come from the builder's seeded scene stream instead of thread_rng().
#include "scenes.hpp"
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <fstream>
#include "../core/rt_math.hpp"
#include "../core/vec3.hpp"

namespace rtx {

using rt::Vec3;
using rt::v3;

// camera.rs:20-57
rt::FlatCamera camera_new(const double lookfrom_[3], const double lookat_[3], const double vup_[3],
                          double vfov, double aspect_ratio, double aperture, double focus_dist,
                          double time1, double time2) {
  Vec3 lookfrom = v3(lookfrom_[0], lookfrom_[1], lookfrom_[2]);
  Vec3 lookat = v3(lookat_[0], lookat_[1], lookat_[2]);
  Vec3 vup = v3(vup_[0], vup_[1], vup_[2]);
  double theta = rt::rt_to_radians(vfov);
  double h = rt::rt_tan(theta / 2.0);
  double viewport_height = 2.0 * h;
  double viewport_width = aspect_ratio * viewport_height;
  Vec3 w = rt::unit(lookfrom - lookat);
  Vec3 u = rt::unit(rt::cross(vup, w));
  Vec3 v = rt::cross(w, u);
  rt::FlatCamera c;
  c.origin = lookfrom;
  c.horizontal = (focus_dist * viewport_width) * u;
  c.vertical = (focus_dist * viewport_height) * v;
  c.lower_left_corner = c.origin - c.horizontal / 2.0 - c.vertical / 2.0 - focus_dist * w;
  c.u = u; c.v = v; c.w = w;
  c.lens_radius = aperture / 2.0;
  c.time1 = time1; c.time2 = time2;
  return c;
}

namespace {

struct S {  // small construction helper over a SceneGraph
  SceneGraph& g;
  explicit S(SceneGraph& gg) : g(gg) {}
  double rnd() { return rt::host_rng_f64(g.rng); }
  double rnd(double lo, double hi) { return rt::host_rng_range(g.rng, lo, hi); }
  Vec3 rnd3() { double x = rnd(), y = rnd(), z = rnd(); return v3(x, y, z); }                            // vec3.rs:273
  Vec3 rnd3(double lo, double hi) { double x = rnd(lo, hi), y = rnd(lo, hi), z = rnd(lo, hi); return v3(x, y, z); }  // vec3.rs:278
  int32_t solid(double r, double gg, double b) { double c[3] = {r, gg, b}; return g.solid_color(c); }
  int32_t lamb(double r, double gg, double b) { return g.lambertian(solid(r, gg, b)); }
  int32_t lamb(Vec3 c) { return lamb(c.x, c.y, c.z); }
  int32_t metal(Vec3 c, double fuzz) { double a[3] = {c.x, c.y, c.z}; return g.metal(a, fuzz); }
  int32_t light(double r, double gg, double b) { return g.diffuse_light(solid(r, gg, b)); }
  int32_t sphere(Vec3 c, double r, int32_t m) { double a[3] = {c.x, c.y, c.z}; return g.sphere(a, r, m); }
  int32_t msphere(Vec3 c0, Vec3 c1, double t0, double t1, double r, int32_t m) {
    double a[3] = {c0.x, c0.y, c0.z}, b[3] = {c1.x, c1.y, c1.z};
    return g.moving_sphere(a, b, t0, t1, r, m);
  }
  int32_t prism(Vec3 p0, Vec3 p1, int32_t m) {
    double a[3] = {p0.x, p0.y, p0.z}, b[3] = {p1.x, p1.y, p1.z};
    return g.rect_prism(a, b, m);
  }
  int32_t translate(Vec3 o, int32_t h) { double a[3] = {o.x, o.y, o.z}; return g.translate(a, h); }
  int32_t medium(Vec3 c, double d, int32_t h) { double a[3] = {c.x, c.y, c.z}; return g.constant_medium(a, d, h); }
  int32_t tri(Vec3 a, Vec3 b, Vec3 c, int32_t m) {
    double x[3] = {a.x, a.y, a.z}, y[3] = {b.x, b.y, b.z}, z[3] = {c.x, c.y, c.z};
    return g.triangle(x, y, z, m);
  }
  int32_t checker_ground() {  // world.rs:98-101
    return g.lambertian(g.checker(solid(0.2, 0.3, 0.1), solid(0.9, 0.9, 0.9)));
  }
};

// world.rs:95-167 (gen_random_scene at HEAD).
int32_t gen_random_scene(S& s) {
  int32_t list = s.g.list_new();
  s.g.list_add(list, s.sphere(v3(0, -1000, -1), 1000.0, s.checker_ground()));
  for (int a = -11; a < 11; ++a) {
    for (int b = -11; b < 11; ++b) {
      double choose_mat = s.rnd();
      double cx = (double)a + 0.9 * s.rnd();
      double cz = (double)b + 0.9 * s.rnd();
      Vec3 center = v3(cx, 0.2, cz);
      if (rt::length(center - v3(4, 0.2, 0)) > 0.9) {
        int32_t mat;
        if (choose_mat < 0.3) {
          Vec3 r1 = s.rnd3(); Vec3 r2 = s.rnd3();
          mat = s.lamb(r1 * r2);
        } else if (choose_mat < 0.6) {
          Vec3 albedo = s.rnd3(0.5, 1.0);
          double fuzz = s.rnd(0.0, 0.5);
          mat = s.metal(albedo, fuzz);
        } else {
          mat = s.g.dielectric(1.5);
        }
        if (choose_mat < 0.8) {
          Vec3 center2 = center + v3(0, 5, 0);
          s.g.list_add(list, s.msphere(center, center2, 0.0, 10.0, 0.2, mat));
          continue;
        }
        s.g.list_add(list, s.sphere(center, 0.2, mat));
      }
    }
  }
  s.g.list_add(list, s.sphere(v3(0, 1, 0), 1.0, s.g.dielectric(1.5)));
  s.g.list_add(list, s.sphere(v3(-4, 1, 0), 1.0, s.lamb(0.4, 0.2, 0.1)));
  s.g.list_add(list, s.sphere(v3(4, 1, 0), 1.0, s.metal(v3(0.7, 0.6, 0.5), 0.0)));
  return s.g.bvh_from_list(list, 0.0, 10.0);
}

// world.rs:169-244 (gen_random_scene_moving): the bouncing-ball scene of the video experiment.
int32_t gen_random_scene_moving(S& s) {
  const double max_time = 100.0;
  int32_t list = s.g.list_new();
  s.g.list_add(list, s.sphere(v3(0, -1000, -1), 1000.0, s.lamb(0.8, 0.8, 0.8)));
  for (int a = -11; a < 11; ++a) {
    for (int b = -11; b < 11; ++b) {
      if (std::abs(a - 0) <= 1 && std::abs(b - 0) <= 1) continue;
      if (std::abs(a - 4) <= 1 && std::abs(b - 0) <= 1) continue;
      double choose_mat = s.rnd();
      double cx = (double)a + 0.9 * s.rnd();
      double cy = 1.7 + s.rnd(0.0, 2.0);
      double cz = (double)b + 0.9 * s.rnd();
      Vec3 center = v3(cx, cy, cz);
      if (rt::length(center - v3(4, 0.2, 0)) > 0.9) {
        int32_t mat;
        if (choose_mat < 0.3) {
          Vec3 r1 = s.rnd3(); Vec3 r2 = s.rnd3();
          mat = s.lamb(r1 * r2);
        } else if (choose_mat < 0.6) {
          Vec3 albedo = s.rnd3(0.5, 1.0);
          double fuzz = s.rnd(0.0, 0.5);
          mat = s.metal(albedo, fuzz);
        } else {
          mat = s.g.dielectric(1.5);
        }
        // world.rs:209-217: `if choose_mat < 1.0` -- always (gen::<f64>() < 1): every small sphere is a GravitySphere
        double c[3] = {center.x, center.y, center.z};
        s.g.list_add(list, s.g.gravity_sphere(c, 0.0, 0.2, mat));
      }
    }
  }
  s.g.list_add(list, s.sphere(v3(0, 1, 0), 1.0, s.g.dielectric(1.5)));
  s.g.list_add(list, s.sphere(v3(-4, 1, 0), 1.0, s.lamb(0.4, 0.2, 0.1)));
  s.g.list_add(list, s.sphere(v3(4, 1, 0), 1.0, s.metal(v3(0.7, 0.6, 0.5), 0.0)));
  return s.g.bvh_from_list(list, 0.0, max_time);
}

// The Book-1 final scene as benchmarked in README.md:12-23 / images/book1.png: the loop
// skeleton, rejection test and big spheres of world.rs:107-116,150-160 with the book's
// material split (0.8 / 0.95), a grey ground and static spheres.
int32_t gen_book1_canonical(S& s) {
  int32_t list = s.g.list_new();
  s.g.list_add(list, s.sphere(v3(0, -1000, 0), 1000.0, s.lamb(0.5, 0.5, 0.5)));
  for (int a = -11; a < 11; ++a) {
    for (int b = -11; b < 11; ++b) {
      double choose_mat = s.rnd();
      double cx = (double)a + 0.9 * s.rnd();
      double cz = (double)b + 0.9 * s.rnd();
      Vec3 center = v3(cx, 0.2, cz);
      if (rt::length(center - v3(4, 0.2, 0)) > 0.9) {
        int32_t mat;
        if (choose_mat < 0.8) {
          Vec3 r1 = s.rnd3(); Vec3 r2 = s.rnd3();
          mat = s.lamb(r1 * r2);
        } else if (choose_mat < 0.95) {
          Vec3 albedo = s.rnd3(0.5, 1.0);
          double fuzz = s.rnd(0.0, 0.5);
          mat = s.metal(albedo, fuzz);
        } else {
          mat = s.g.dielectric(1.5);
        }
        s.g.list_add(list, s.sphere(center, 0.2, mat));
      }
    }
  }
  s.g.list_add(list, s.sphere(v3(0, 1, 0), 1.0, s.g.dielectric(1.5)));
  s.g.list_add(list, s.sphere(v3(-4, 1, 0), 1.0, s.lamb(0.4, 0.2, 0.1)));
  s.g.list_add(list, s.sphere(v3(4, 1, 0), 1.0, s.metal(v3(0.7, 0.6, 0.5), 0.0)));
  return s.g.bvh_from_list(list, 0.0, 1.0);
}

// world.rs:246-265
int32_t gen_checkered_sphere(S& s) {
  int32_t list = s.g.list_new();
  int32_t ground = s.checker_ground();
  s.g.list_add(list, s.sphere(v3(0, -10, 0), 10.0, ground));
  s.g.list_add(list, s.sphere(v3(0, 10, 0), 10.0, ground));
  return list;
}

// world.rs:267-285
int32_t gen_two_perlin(S& s) {
  int32_t list = s.g.list_new();
  int32_t ground = s.g.lambertian(s.g.noise(4.0));
  s.g.list_add(list, s.sphere(v3(0, -1000, 0), 1000.0, ground));
  s.g.list_add(list, s.sphere(v3(0, 2, 0), 2.0, ground));
  return list;
}

int32_t earth_texture(S& s, const SceneOptions& opt) {
  if (opt.earth_ppm) {
    std::ifstream probe(opt.earth_ppm);
    if (probe.good()) return s.g.image_from_ppm(opt.earth_ppm);
  }
  std::vector<double> tx;
  procedural_earth(1024, 512, &tx);
  return s.g.image_from_texels(1024, 512, tx.data());
}

// world.rs:287-305
int32_t earth(S& s, const SceneOptions& opt) {
  int32_t list = s.g.list_new();
  int32_t tex = earth_texture(s, opt);
  if (tex < 0) return -1;
  int32_t ground = s.g.lambertian(tex);
  s.g.list_add(list, s.sphere(v3(0, -1000, 0), 1000.0, ground));
  s.g.list_add(list, s.sphere(v3(0, 2, 0), 2.0, ground));
  return list;
}

// world.rs:307-342
int32_t gen_simple_light(S& s) {
  int32_t list = s.g.list_new();
  int32_t ground = s.g.lambertian(s.g.noise(4.0));
  s.g.list_add(list, s.sphere(v3(0, -1000, 0), 1000.0, ground));
  s.g.list_add(list, s.sphere(v3(0, 2, 0), 2.0, ground));
  int32_t difflight = s.light(10, 10, 10);
  s.g.list_add(list, s.g.rect(H_XY_RECT, 3.0, 5.0, 1.0, 3.0, -2.0, difflight));
  s.g.list_add(list, s.sphere(v3(0, 10, 0), 3.0, difflight));
  return list;
}

// The six walls + light shared by world.rs:344-386, 415-457, 753-795.
void cornell_walls(S& s, int32_t list, int32_t* white_out) {
  int32_t red = s.lamb(0.65, 0.05, 0.05);
  int32_t white = s.lamb(0.73, 0.73, 0.73);
  int32_t green = s.lamb(0.12, 0.45, 0.15);
  int32_t light = s.light(15, 15, 15);
  s.g.list_add(list, s.g.rect(H_YZ_RECT, 0.0, 555.0, 0.0, 555.0, 555.0, green));
  s.g.list_add(list, s.g.rect(H_YZ_RECT, 0.0, 555.0, 0.0, 555.0, 0.0, red));
  s.g.list_add(list, s.g.rect(H_XZ_RECT, 213.0, 343.0, 227.0, 332.0, 554.0, light));
  s.g.list_add(list, s.g.rect(H_XZ_RECT, 0.0, 555.0, 0.0, 555.0, 0.0, white));
  s.g.list_add(list, s.g.rect(H_XZ_RECT, 0.0, 555.0, 0.0, 555.0, 555.0, white));
  s.g.list_add(list, s.g.rect(H_XY_RECT, 0.0, 555.0, 0.0, 555.0, 555.0, white));
  *white_out = white;
}

// world.rs:344-413
int32_t cornell_box(S& s) {
  int32_t list = s.g.list_new();
  int32_t white;
  cornell_walls(s, list, &white);
  s.g.list_add(list, s.translate(v3(265, 0, 295), s.g.rotate_y(15.0, s.prism(v3(0, 0, 0), v3(165, 330, 165), white))));
  s.g.list_add(list, s.translate(v3(130, 0, 65), s.g.rotate_y(-18.0, s.prism(v3(0, 0, 0), v3(165, 165, 165), white))));
  return list;
}

// world.rs:415-492
int32_t cornell_smoke(S& s) {
  int32_t list = s.g.list_new();
  int32_t white;
  cornell_walls(s, list, &white);
  s.g.list_add(list, s.medium(v3(0, 0, 0), 0.01,
                              s.translate(v3(265, 0, 295), s.g.rotate_y(15.0, s.prism(v3(0, 0, 0), v3(165, 330, 165), white)))));
  s.g.list_add(list, s.medium(v3(1, 1, 1), 0.01,
                              s.translate(v3(130, 0, 65), s.g.rotate_y(-18.0, s.prism(v3(0, 0, 0), v3(165, 165, 165), white)))));
  return list;
}

// world.rs:494-616
int32_t final_scene(S& s, const SceneOptions& opt) {
  int32_t list = s.g.list_new();
  int32_t boxes1 = s.g.list_new();
  int32_t ground = s.g.lambertian(s.solid(0.48, 0.83, 0.53));
  int boxes_per_side = opt.book2_boxes_per_side;
  for (int ii = 0; ii < boxes_per_side; ++ii) {
    for (int jj = 0; jj < boxes_per_side; ++jj) {
      double i = (double)ii, j = (double)jj;
      double w = 100.0;
      double x0 = -1000.0 + i * w;
      double z0 = -1000.0 + j * w;
      double y0 = 0.0;
      double x1 = x0 + w;
      double y1 = s.rnd(1.0, 101.0);
      double z1 = z0 + w;
      s.g.list_add(boxes1, s.prism(v3(x0, y0, z0), v3(x1, y1, z1), ground));
    }
  }
  s.g.list_add(list, s.g.bvh_from_list(boxes1, 0.0, 1.0));
  s.g.list_add(list, s.g.rect(H_XZ_RECT, 123.0, 432.0, 147.0, 412.0, 554.0, s.light(7, 7, 7)));
  Vec3 center1 = v3(400, 400, 400);
  Vec3 center2 = center1 + v3(30, 0, 0);
  s.g.list_add(list, s.msphere(center1, center2, 0.0, 1.0, 50.0, s.lamb(0.7, 0.3, 1)));
  s.g.list_add(list, s.sphere(v3(260, 150, 45), 50.0, s.g.dielectric(1.5)));
  s.g.list_add(list, s.sphere(v3(0, 150, 145), 50.0, s.metal(v3(0.8, 0.8, 0.9), 1.0)));
  s.g.list_add(list, s.sphere(v3(360, 150, 145), 70.0, s.g.dielectric(1.5)));
  s.g.list_add(list, s.medium(v3(0.2, 0.4, 0.9), 0.2, s.sphere(v3(360, 150, 145), 70.0, s.g.dielectric(1.5))));
  s.g.list_add(list, s.sphere(v3(0, 0, 0), 5000.0, s.g.dielectric(1.5)));
  s.g.list_add(list, s.medium(v3(1, 1, 1), 0.0001, s.sphere(v3(0, 0, 0), 5000.0, s.g.dielectric(1.5))));
  int32_t tex = earth_texture(s, opt);
  if (tex < 0) return -1;
  s.g.list_add(list, s.sphere(v3(400, 200, 400), 100.0, s.g.lambertian(tex)));
  s.g.list_add(list, s.sphere(v3(220, 280, 300), 80.0, s.g.lambertian(s.g.noise(0.1))));
  int32_t white = s.lamb(0.73, 0.73, 0.73);
  int32_t boxes2 = s.g.list_new();
  for (int n = 0; n < opt.book2_spheres; ++n) s.g.list_add(boxes2, s.sphere(s.rnd3(0.0, 165.0), 10.0, white));
  s.g.list_add(list, s.translate(v3(-100, 270, 395), s.g.rotate_y(15.0, s.g.bvh_from_list(boxes2, 0.0, 1.0))));
  return list;
}

// world.rs:618-647
int32_t gen_moving_test(S& s) {
  int32_t list = s.g.list_new();
  s.g.list_add(list, s.sphere(v3(0, -1000, -1), 1000.0, s.checker_ground()));
  s.g.list_add(list, s.msphere(v3(2, -1, 2), v3(2, 7, 2), 0.0, 10.0, 1.0, s.lamb(1, 0, 0)));
  return s.g.bvh_from_list(list, 0.0, 10.0);
}

// world.rs:649-663: one sphere wrapped in 20 nested single-element lists.
int32_t benchmark_test_scene(S& s) {
  int32_t amit = s.g.list_new();
  s.g.list_add(amit, s.sphere(v3(0, 0, 0), 4.0, s.lamb(0.5, 0.5, 0.5)));
  for (int i = 0; i < 19; ++i) {
    int32_t tramit = s.g.list_new();
    s.g.list_add(tramit, amit);
    amit = tramit;
  }
  return amit;
}

// world.rs:665-679
int32_t triangle_test(S& s) {
  int32_t list = s.g.list_new();
  s.g.list_add(list, s.tri(v3(0, 5, 0), v3(5, 0, 0), v3(0, 0, 0), s.lamb(1, 0, 0)));
  s.g.list_add(list, s.sphere(v3(5, 0, 0), 1.0, s.lamb(0, 1, 0)));
  return list;
}

// world.rs:681-751
int32_t stanford_dragon(S& s, const SceneOptions& opt) {
  int32_t list = s.g.list_new();
  int32_t dragon_list = -1;
  if (opt.dragon_ply) {
    std::ifstream probe(opt.dragon_ply);
    if (probe.good()) {
      dragon_list = s.g.triangle_model(opt.dragon_ply, 100.0);
      if (dragon_list < 0) return -1;
    }
  }
  if (dragon_list < 0) {
    std::vector<double> verts;
    std::vector<int64_t> faces;
    procedural_mesh(opt.mesh_triangles, &verts, &faces);
    dragon_list = s.g.triangle_mesh(verts.data(), (int64_t)verts.size() / 3, faces.data(),
                                    (int64_t)faces.size() / 3, s.lamb(0.2, 0.2, 0.2));
    if (dragon_list < 0) return -1;
  }
  int32_t dragon = s.g.bvh_from_list(dragon_list, 0.0, 1.0);
  if (dragon < 0) return -1;
  int32_t light = s.light(4, 4, 4);
  int32_t backdrop = s.g.rect(H_XY_RECT, -100.0, 100.0, -100.0, 100.0, -20.0, s.lamb(0.8, 0.3, 0.3));
  int32_t backwall = s.g.rect(H_XY_RECT, -100.0, 100.0, -100.0, 100.0, 20.0, s.lamb(1, 1, 1));
  int32_t ground = s.g.rect(H_XZ_RECT, -40.0, 40.0, -40.0, 40.0, 5.0, s.metal(v3(0.3, 0.3, 0.3), 0.02));
  int32_t ceiling = s.g.rect(H_XZ_RECT, -100.0, 100.0, -100.0, 100.0, 55.0, s.metal(v3(1, 1, 1), 0.0));
  int32_t left_wall = s.g.rect(H_YZ_RECT, -100.0, 100.0, -100.0, 100.0, -30.0, s.lamb(0.3, 0.8, 0.3));
  int32_t right_wall = s.g.rect(H_YZ_RECT, -100.0, 100.0, -100.0, 100.0, 30.0, s.lamb(0.3, 0.3, 0.8));
  int32_t ceiling_light = s.g.rect(H_XZ_RECT, -100.0, 100.0, -100.0, 100.0, 55.0, light);
  s.g.list_add(list, dragon);
  s.g.list_add(list, backdrop);
  s.g.list_add(list, backwall);
  s.g.list_add(list, ground);
  s.g.list_add(list, ceiling);
  s.g.list_add(list, left_wall);
  s.g.list_add(list, right_wall);
  s.g.list_add(list, ceiling_light);
  return list;
}

// world.rs:753-874
int32_t triangular_prism(S& s) {
  int32_t list = s.g.list_new();
  int32_t white;
  cornell_walls(s, list, &white);
  s.g.list_add(list, s.tri(v3(200, 0, 200), v3(300, 0, 200), v3(250, 250, 200), white));
  s.g.list_add(list, s.g.rect(H_XY_RECT, 0.0, 300.0, 0.0, 150.0, 201.0, white));
  return list;
}

}  // namespace

bool get_world_cam(SceneGraph& g, int32_t scene_id, const SceneOptions& opt, WorldCam* out,
                   std::string* err) {
  S s(g);
  double aspect = 16.0 / 9.0;  // world.rs:878
  double bg[3] = {0.7, 0.8, 1.0};  // world.rs:879
  double lookfrom[3] = {13, 2, 3}, lookat[3] = {0, 0, 0}, vup[3] = {0, 1, 0};
  double vfov = 20.0, aperture = 0.0, dist_to_focus = 10.0, t1 = 0.0, t2 = 1.0;
  int32_t world = -1;
  auto set3 = [](double* d, double a, double b, double c) { d[0] = a; d[1] = b; d[2] = c; };
  switch (scene_id) {
    case RTX_SCENE_CHECKERED_SPHERES: world = gen_checkered_sphere(s); break;  // world.rs:881-901
    case RTX_SCENE_TWO_PERLIN: world = gen_two_perlin(s); break;               // world.rs:902-922
    case RTX_SCENE_EARTH: world = earth(s, opt); break;                        // world.rs:923-943
    case RTX_SCENE_SIMPLE_LIGHT:                                               // world.rs:945-966
      world = gen_simple_light(s);
      set3(lookfrom, 26, 3, 6); set3(lookat, 0, 2, 0); set3(bg, 0, 0, 0);
      break;
    case RTX_SCENE_CORNELL_BOX:                                                // world.rs:967-987
    case RTX_SCENE_CORNELL_SMOKE:                                              // world.rs:988-1008
    case RTX_SCENE_TRIANGULAR_PRISM:                                           // world.rs:1136-1156
      world = scene_id == RTX_SCENE_CORNELL_BOX ? cornell_box(s)
              : scene_id == RTX_SCENE_CORNELL_SMOKE ? cornell_smoke(s) : triangular_prism(s);
      set3(lookfrom, 278, 278, -800); set3(lookat, 278, 278, 0); set3(bg, 0, 0, 0);
      vfov = 40.0; aspect = 1.0;
      break;
    case RTX_SCENE_BOOK2_FINAL:                                                // world.rs:1009-1029
      world = final_scene(s, opt);
      set3(lookfrom, 478, 278, -600); set3(lookat, 278, 278, 0); set3(bg, 0, 0, 0);
      vfov = 40.0; aspect = 1.0;
      break;
    case RTX_SCENE_MOVING_TEST:                                                // world.rs:1030-1050
      world = gen_moving_test(s);
      aperture = 0.1; t1 = 2.0; t2 = 2.5;
      break;
    case RTX_SCENE_RANDOM_MOVING:                                              // world.rs:1051-1071
      world = gen_random_scene_moving(s);
      aperture = 0.1; t2 = 10.0;
      break;
    case RTX_SCENE_BENCHMARK_TEST:                                             // world.rs:1072-1092
      world = benchmark_test_scene(s);
      aperture = 0.1; t2 = 10.0;
      break;
    case RTX_SCENE_TRIANGLE_TEST:                                              // world.rs:1093-1113
      world = triangle_test(s);
      set3(lookfrom, 0, 0, 20); aperture = 0.1; t2 = 10.0;
      break;
    case RTX_SCENE_STANFORD_DRAGON:                                            // world.rs:1114-1134
      world = stanford_dragon(s, opt);
      set3(lookfrom, 0, 20, 20); set3(lookat, 0, 11, 0);
      vfov = 60.0; dist_to_focus = 40.0; t2 = 10.0;
      break;
    case RTX_SCENE_BOOK1_CANONICAL:
      world = gen_book1_canonical(s);
      aperture = 0.1; aspect = 3.0 / 2.0;  // README.md:12
      break;
    case RTX_SCENE_EMPTY:
      world = g.list_new();
      break;
    default:                                                                   // world.rs:1157-1177
      world = gen_random_scene(s);
      aperture = 0.1; t2 = 10.0;
      break;
  }
  if (world < 0) { *err = g.error.empty() ? "scene construction failed" : g.error; return false; }
  if (opt.camera_aspect > 0.0) aspect = opt.camera_aspect;
  out->world = world;
  out->cam = camera_new(lookfrom, lookat, vup, vfov, aspect, aperture, dist_to_focus, t1, t2);
  out->background[0] = bg[0]; out->background[1] = bg[1]; out->background[2] = bg[2];
  out->camera_aspect = aspect;
  return true;
}

// ------------------------------------------------------------------ procedural assets
// Stand-in for "earthshit.ppm" (world.rs:290,580; not shipped): a lat-long map with
// continent-like blobs from a few sinusoids, integer 0..255 channels like a real P3 file.
void procedural_earth(int32_t w, int32_t h, std::vector<double>* texels) {
  texels->resize((size_t)3 * w * h);
  for (int32_t j = 0; j < h; ++j) {
    for (int32_t i = 0; i < w; ++i) {
      double lon = 2.0 * RT_PI * ((double)i + 0.5) / (double)w;
      double lat = RT_PI * ((double)j + 0.5) / (double)h - RT_PI / 2.0;
      double f = rt::rt_sin(3.0 * lon) * rt::rt_cos(2.0 * lat) + 0.5 * rt::rt_sin(7.0 * lon + 1.3) * rt::rt_sin(5.0 * lat) +
                 0.25 * rt::rt_cos(13.0 * lon) * rt::rt_cos(11.0 * lat + 0.7);
      double ice = rt::rt_fabs(lat) > 1.25 ? 1.0 : 0.0;
      double r, g, b;
      if (ice > 0.0) { r = 235; g = 240; b = 245; }
      else if (f > 0.15) { r = 60 + 90 * (f - 0.15); g = 140 - 40 * (f - 0.15); b = 50; }
      else { r = 20; g = 60 + 40 * (f + 1.5) / 1.65; b = 150 + 60 * (f + 1.5) / 1.65; }
      size_t o = 3 * ((size_t)j * w + i);
      (*texels)[o] = std::floor(std::fmin(255.0, std::fmax(0.0, r)));
      (*texels)[o + 1] = std::floor(std::fmin(255.0, std::fmax(0.0, g)));
      (*texels)[o + 2] = std::floor(std::fmin(255.0, std::fmax(0.0, b)));
    }
  }
}

// Stand-in for dragon_vrip*.ply (world.rs:684; models/ is git-ignored upstream): a closed,
// bumpy torus-like surface tessellated into ~target triangles and placed where the scaled
// dragon sits (x in [-11.1, 9.5], y in [5.3, 19.8], z in [-5.0, 4.1]).
void procedural_mesh(int64_t target_triangles, std::vector<double>* vertices,
                     std::vector<int64_t>* faces) {
  if (target_triangles < 8) target_triangles = 8;
  // nu : nv = 9 : 4, 2 triangles per quad.
  int64_t nv = (int64_t)std::floor(std::sqrt((double)target_triangles / 2.0 * 4.0 / 9.0));
  if (nv < 2) nv = 2;
  int64_t nu = target_triangles / (2 * nv);
  if (nu < 2) nu = 2;
  vertices->resize((size_t)3 * nu * nv);
  for (int64_t a = 0; a < nu; ++a) {
    double u = 2.0 * RT_PI * (double)a / (double)nu;
    for (int64_t b = 0; b < nv; ++b) {
      double v = 2.0 * RT_PI * (double)b / (double)nv;
      double R = 1.0 + 0.15 * rt::rt_sin(3.0 * u);
      double r = 0.42 + 0.08 * rt::rt_sin(5.0 * u + 2.0 * v) + 0.04 * rt::rt_cos(17.0 * v + 11.0 * u);
      double x = (R + r * rt::rt_cos(v)) * rt::rt_cos(u);
      double z = (R + r * rt::rt_cos(v)) * rt::rt_sin(u);
      double y = r * rt::rt_sin(v) + 0.25 * rt::rt_sin(2.0 * u);
      // map to the dragon's extent: x by 6.6, y to [5.3, 19.8], z by 2.9
      size_t o = 3 * (size_t)(a * nv + b);
      (*vertices)[o] = -0.8 + 6.6 * x;
      (*vertices)[o + 1] = 12.55 + 9.0 * y;
      (*vertices)[o + 2] = -0.45 + 2.9 * z;
    }
  }
  faces->clear();
  faces->reserve((size_t)6 * nu * nv);
  for (int64_t a = 0; a < nu; ++a) {
    int64_t a1 = (a + 1) % nu;
    for (int64_t b = 0; b < nv; ++b) {
      int64_t b1 = (b + 1) % nv;
      int64_t p00 = a * nv + b, p10 = a1 * nv + b, p01 = a * nv + b1, p11 = a1 * nv + b1;
      faces->push_back(p00); faces->push_back(p10); faces->push_back(p11);
      faces->push_back(p00); faces->push_back(p11); faces->push_back(p01);
    }
  }
}

}  // namespace rtx
