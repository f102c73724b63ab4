// Vec3 / Ray arithmetic with the reference's operation order.
//
// Follows /root/reference/src/vec3.rs:35-121 (methods) and 139-251 (operators),
// /root/reference/src/ray.rs:11-33 and /root/reference/src/mutil.rs:1-9.
// Each operator performs exactly the per-component IEEE operation the Rust
// operator impl performs, in the same operand order, so host (g++) and device
// (hipcc) evaluate bit-identical results under -ffp-contract=off.
#pragma once
#include "rt_config.hpp"

namespace rt {

struct Vec3 {
  real x, y, z;
};
typedef Vec3 Point3;
typedef Vec3 Color;

RT_HD Vec3 v3(real x, real y, real z) { Vec3 v; v.x = x; v.y = y; v.z = z; return v; }

// vec3.rs:199-209 (Add), 211-221 (Sub)
RT_HD Vec3 operator+(Vec3 a, Vec3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_HD Vec3 operator-(Vec3 a, Vec3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
// vec3.rs:139-149 (Vec3 * Vec3), 163-173 (Vec3 * T), 175-185 (f64 * Vec3)
RT_HD Vec3 operator*(Vec3 a, Vec3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
RT_HD Vec3 operator*(Vec3 a, real t) { return v3(a.x * t, a.y * t, a.z * t); }
RT_HD Vec3 operator*(real t, Vec3 a) { return v3(t * a.x, t * a.y, t * a.z); }
// vec3.rs:187-197 (Vec3 / T): three true divisions, not a reciprocal multiply.
RT_HD Vec3 operator/(Vec3 a, real t) { return v3(a.x / t, a.y / t, a.z / t); }
// vec3.rs:151-161 (Neg): multiply by -1.0 (exact sign flip).
RT_HD Vec3 operator-(Vec3 a) { return v3(a.x * -real(1.0), a.y * -real(1.0), a.z * -real(1.0)); }
// vec3.rs:223-229 (AddAssign), 245-251 (MulAssign<Vec3>)
RT_HD void operator+=(Vec3& a, Vec3 b) { a.x += b.x; a.y += b.y; a.z += b.z; }
RT_HD void operator*=(Vec3& a, Vec3 b) { a.x *= b.x; a.y *= b.y; a.z *= b.z; }
// vec3.rs:231-237 (MulAssign<T>), 239-243 (DivAssign<T> = multiply by 1/rhs)
RT_HD void operator*=(Vec3& a, real t) { a.x *= t; a.y *= t; a.z *= t; }
RT_HD void operator/=(Vec3& a, real t) { a *= (real(1.0) / t); }
// vec3.rs:253-259
RT_HD bool operator==(Vec3 a, Vec3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }

// vec3.rs:43-45: x*x' + y*y' + z*z' evaluated left to right.
RT_HD real dot(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// vec3.rs:47-53
RT_HD Vec3 cross(Vec3 a, Vec3 b) {
  return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// vec3.rs:35-41
RT_HD real length_squared(Vec3 a) { return dot(a, a); }
RT_HD real length(Vec3 a) { return rt_sqrt(length_squared(a)); }
// vec3.rs:55-57: division by the length (three divides).
RT_HD Vec3 unit(Vec3 a) { return a / length(a); }
// vec3.rs:59-62
RT_HD bool near_zero(Vec3 a) {
  const real s = real(1e-8);
  return rt_fabs(a.x) < s && rt_fabs(a.y) < s && rt_fabs(a.z) < s;
}
// vec3.rs:64-66: *self - 2.0 * self.dot(normal) * *normal  ==  v - ((2.0*d) * n)
RT_HD Vec3 reflect(Vec3 v, Vec3 n) { return v - (real(2.0) * dot(v, n)) * n; }
// vec3.rs:116-121
RT_HD Vec3 refract(Vec3 uv, Vec3 n, real etai_over_etat) {
  real cos_theta = rt_fmin(dot(-uv, n), real(1.0));
  Vec3 r_out_perp = etai_over_etat * (uv + cos_theta * n);
  Vec3 r_out_parallel = (-(rt_sqrt(rt_fabs(real(1.0) - length_squared(r_out_perp))))) * n;
  return r_out_perp + r_out_parallel;
}

// mutil.rs:1-9 (NaN falls through both comparisons and is returned unchanged).
RT_HD real clamp(real x, real mn, real mx) {
  if (x < mn) return mn;
  if (x > mx) return mx;
  return x;
}

// ray.rs:3-33
struct Ray {
  Point3 origin;
  Vec3 direction;
  real time;
};
RT_HD Ray make_ray(Point3 o, Vec3 d, real time) { Ray r; r.origin = o; r.direction = d; r.time = time; return r; }
// ray.rs:31-33: origin + direction * t
RT_HD Point3 ray_at(const Ray& r, real t) { return r.origin + r.direction * t; }

// vec3.rs:89-107 (get_normalized_color): tone map one accumulated pixel to the
// three integer channel values.  scale = 1/spp, multiply, sqrt, clamp, *255.9,
// saturating cast (NaN -> 0).
RT_HD void tone_map(Color sum, uint32_t samples_per_pixel, int32_t out[3]) {
  const real COLOR_MAX = real(255.9);  // vec3.rs:10
  real scale = real(1.0) / (real)samples_per_pixel;
  real r = sum.x * scale, g = sum.y * scale, b = sum.z * scale;
  r = rt_sqrt(r); g = rt_sqrt(g); b = rt_sqrt(b);
  out[0] = rt_f64_as_i32(COLOR_MAX * clamp(r, real(0.0), real(1.0)));
  out[1] = rt_f64_as_i32(COLOR_MAX * clamp(g, real(0.0), real(1.0)));
  out[2] = rt_f64_as_i32(COLOR_MAX * clamp(b, real(0.0), real(1.0)));
}

}  // namespace rt
