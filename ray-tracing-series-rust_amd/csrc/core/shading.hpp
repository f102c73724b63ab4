// Samplers, textures, Perlin noise and material scatter -- the L2 half of the hot path.
//
// Reference: /root/reference/src/vec3.rs:273-322 (samplers), texture.rs (textures),
// perlin.rs:28-106 (noise), hit.rs:992-1152 (materials).  Draw order from the
// path's stream is the reference's draw order.
#pragma once
#include "geometry.hpp"

namespace rt {

// vec3.rs:273-276
RT_HD Vec3 random_vec3(Rng& g) {
  real x = rng_f64(g), y = rng_f64(g), z = rng_f64(g);
  return v3(x, y, z);
}
// vec3.rs:278-285
RT_HD Vec3 random_range_vec3(Rng& g, real mn, real mx) {
  real x = rng_range(g, mn, mx), y = rng_range(g, mn, mx), z = rng_range(g, mn, mx);
  return v3(x, y, z);
}
// One iteration of random_in_unit_sphere's rejection loop (vec3.rs:288-294): three draws, accept test.
RT_HD bool sphere_sample_try(Rng& g, Vec3* p) {
  real x = rng_range_pm1(g), y = rng_range_pm1(g), z = rng_range_pm1(g);  // Vec3::random_range(-1.0, 1.0)
  *p = v3(x, y, z);
  return length_squared(*p) < real(1.0);
}
// vec3.rs:287-295
RT_HD Vec3 random_in_unit_sphere(Rng& g) {
  for (;;) {
    Vec3 p;
    if (sphere_sample_try(g, &p)) return p;
  }
}
// vec3.rs:297-299
RT_HD Vec3 random_unit_vector(Rng& g) { return unit(random_in_unit_sphere(g)); }
// vec3.rs:310-322
RT_HD Vec3 random_in_unit_disk(Rng& g) {
  for (;;) {
    real x = rng_range_pm1(g), y = rng_range_pm1(g);
    Vec3 p = v3(x, y, real(0.0));
    if (length_squared(p) < real(1.0)) return p;
  }
}

// camera.rs:59-71.  The lens sample and the shutter time are drawn even when
// aperture is 0 / the shutter interval is a single instant's worth.
RT_HD Ray camera_get_ray(const FlatCamera& c, real s, real t, Rng& g) {
  Vec3 rd = c.lens_radius * random_in_unit_disk(g);
  Vec3 offset = c.u * rd.x + c.v * rd.y;
  Point3 origin = c.origin + offset;
  Vec3 direction = c.lower_left_corner + s * c.horizontal + t * c.vertical - c.origin - offset;
  real time = rng_range(g, c.time1, c.time2);
  return make_ray(origin, direction, time);
}

// perlin.rs:28-52 + 85-106 (gradient fetch fused into the interpolation loop; same values).
template <bool COUNT>
RT_HD real perlin_noise(const FlatPerlin& pn, Point3 p, TraceCounters* cnt) {
  if (COUNT) cnt->perlin_calls++;
  real fx = rt_floor(p.x), fy = rt_floor(p.y), fz = rt_floor(p.z);
  real u = p.x - fx, v = p.y - fy, w = p.z - fz;
  uint32_t i = (uint32_t)rt_f64_as_i32(fx), j = (uint32_t)rt_f64_as_i32(fy), k = (uint32_t)rt_f64_as_i32(fz);
  real uu = u * u * (real(3.0) - real(2.0) * u);
  real vv = v * v * (real(3.0) - real(2.0) * v);
  real ww = w * w * (real(3.0) - real(2.0) * w);
  real accum = real(0.0);
  RT_NO_UNROLL
  for (uint32_t di = 0; di < 2; ++di)
    RT_NO_UNROLL
    for (uint32_t dj = 0; dj < 2; ++dj)
      RT_NO_UNROLL
      for (uint32_t dk = 0; dk < 2; ++dk) {
        int32_t h = (int32_t)xor3((uint32_t)pn.perm_x[(i + di) & 255u], (uint32_t)pn.perm_y[(j + dj) & 255u], (uint32_t)pn.perm_z[(k + dk) & 255u]);
        Vec3 c = load_v3(pn.ranvec[h]);
        real i1 = (real)di, j1 = (real)dj, k1 = (real)dk;
        Vec3 weight_v = v3(u - i1, v - j1, w - k1);
        accum += (i1 * uu + (real(1.0) - i1) * (real(1.0) - uu)) * (j1 * vv + (real(1.0) - j1) * (real(1.0) - vv)) *
                 (k1 * ww + (real(1.0) - k1) * (real(1.0) - ww)) * dot(c, weight_v);
      }
  return accum;
}

// perlin.rs:54-66
template <bool COUNT>
RT_HD real perlin_turbulence(const FlatPerlin& pn, Point3 p, int depth, TraceCounters* cnt) {
  real accum = real(0.0);
  Point3 temp_p = p;
  real weight = real(1.0);
  RT_NO_UNROLL
  for (int i = 0; i < depth; ++i) {
    accum += weight * perlin_noise<COUNT>(pn, temp_p, cnt);
    weight *= real(0.5);
    temp_p = temp_p * real(2.0);
  }
  return rt_fabs(accum);
}

// The two expensive, rarely reached texture kinds, kept out of line so that their registers and
// code do not weigh on the bounce loop (Noise: 7 Perlin octaves; Image: a texel fetch).
template <bool COUNT>
RT_HD Color texture_value_cold(const SceneView& sv, const FlatTexture& t, real u, real v, Point3 p,
                                        TraceCounters* cnt) {
  if (t.kind == TEX_NOISE) {  // texture.rs:80-88
    real s = real(1.0) + rt_sin(t.scale * p.z + real(10.0) * perlin_turbulence<COUNT>(sv.perlins[t.a], p, 7, cnt));
    return v3(real(1.0), real(1.0), real(1.0)) * real(0.5) * s;
  }
  // TEX_IMAGE, texture.rs:102-121
  if (COUNT) cnt->texels++;
  const FlatImage& im = sv.images[t.a];
  real uc = clamp(u, real(0.0), real(1.0));
  real vc = real(1.0) - clamp(v, real(0.0), real(1.0));
  int32_t i = rt_f64_as_i32(uc * (real)im.width);
  int32_t j = rt_f64_as_i32(vc * (real)im.height);
  i = i < im.width - 1 ? i : im.width - 1;
  j = j < im.height - 1 ? j : im.height - 1;
  const real color_scale = real(1.0) / real(255.0);
  const real* px = sv.texels + 3 * (im.first_texel + (int64_t)j * im.width + i);
  return v3(color_scale * px[0], color_scale * px[1], color_scale * px[2]);
}

// Texture::value for the whole texture tree (texture.rs:27-31, 54-64, 80-88, 102-121).
// Checker picks a child from p alone, so nested checkers resolve iteratively.
template <uint32_t F, bool COUNT>
RT_HD Color texture_value(const SceneView& sv, int32_t tex, real u, real v, Point3 p,
                          TraceCounters* cnt) {
  if (F & F_CHECKER) {
    for (int guard = 0; guard < 16; ++guard) {
      const FlatTexture& t = sv.textures[tex];
      if (t.kind != TEX_CHECKER) break;
      // texture.rs:56-62: sines = sin(10x) sin(10y) sin(10z); sines < 0 -> odd.  Only the sign is used, and
      // rt_sin_sign gives exactly the sign rt_sin would have (a NaN or a zero factor makes the test false).
      int sx = rt_sin_sign(real(10.0) * p.x), sy = rt_sin_sign(real(10.0) * p.y), sz = rt_sin_sign(real(10.0) * p.z);
      bool negative = sx != 2 && sy != 2 && sz != 2 && sx * sy * sz < 0;
      tex = negative ? t.b : t.a;
    }
  }
  const FlatTexture& t = sv.textures[tex];
  if ((F & (F_NOISE | F_IMAGE)) && (t.kind == TEX_NOISE || t.kind == TEX_IMAGE))
    return texture_value_cold<COUNT>(sv, t, u, v, p, cnt);
  return load_v3(t.color);  // TEX_SOLID
}

// hit.rs:1095-1099;  f64::powi(x, 5) == x * (x*x) * (x*x) evaluated as ((x^2)^2)*x by
// LLVM's powi expansion (square-and-multiply from the low bit: r = x; x2 = x*x; x4 = x2*x2; r*x4).
RT_HD real reflectance(real cosine, real ref_idx) {
  real r0 = (real(1.0) - ref_idx) / (real(1.0) + ref_idx);
  r0 = r0 * r0;
  real b = real(1.0) - cosine;
  real b2 = b * b;
  real b4 = b2 * b2;
  return r0 + (real(1.0) - r0) * (b * b4);
}

// reflectance with r0 * r0 already at hand (the flattener's dielectric_constants below computes it with reflectance's own expressions).
RT_HD real reflectance_r0sq(real cosine, real r0sq) {
  real b = real(1.0) - cosine;
  real b2 = b * b;
  real b4 = b2 * b2;
  return r0sq + (real(1.0) - r0sq) * (b * b4);
}
// What a Dielectric's scatter needs of its index alone: out[0] = 1 / ir (hit.rs:1104-1108), out[1] / out[2] = r0 * r0 of
// reflectance(cos, 1 / ir) / reflectance(cos, ir) (hit.rs:1095-1097).
inline void dielectric_constants(real ir, real out[3]) {
  out[0] = real(1.0) / ir;
  real r0 = (real(1.0) - out[0]) / (real(1.0) + out[0]);
  out[1] = r0 * r0;
  r0 = (real(1.0) - ir) / (real(1.0) + ir);
  out[2] = r0 * r0;
}

// Material::emitted (hit.rs:1015-1017 default, 1149-1151 DiffuseLight).
// Texture::value of a material's texture.  A scene without checker, noise and image textures has SolidColors only, and the
// flattener has copied each one's colour into the material record: no second, dependent fetch.
template <uint32_t F, bool COUNT>
RT_HD Color material_texture_value(const SceneView& sv, const FlatMaterial& m, const HitRecord& rec, TraceCounters* cnt) {
  if (!(F & (F_CHECKER | F_NOISE | F_IMAGE))) return load_v3(m.albedo);
  return texture_value<F, COUNT>(sv, m.tex, rec.u, rec.v, rec.p, cnt);
}

template <uint32_t F, bool COUNT>
RT_HD Color material_emitted(const SceneView& sv, const FlatMaterial& m, const HitRecord& rec,
                             TraceCounters* cnt) {
  if ((F & F_LIGHT) && m.kind == MAT_DIFFUSE_LIGHT)
    return material_texture_value<F, COUNT>(sv, m, rec, cnt);
  return v3(0, 0, 0);
}

// Material::scatter for all five materials.
// Lambertian, Metal and Isotropic each begin by drawing one random_in_unit_sphere() sample
// (hit.rs:1040 via random_unit_vector, hit.rs:1074, hit.rs:1007) and draw nothing else, so scatter is
// split in two: (1) does this material want a sphere sample, (2) the rest of scatter given that
// sample.  material_scatter() glues them in the reference's order; the device may run the rejection
// loop of (1) elsewhere (its own stage / its own loop) as long as the path's draws keep their order.
RT_HD bool material_needs_sphere_sample(int32_t kind) {
  return kind == MAT_LAMBERTIAN || kind == MAT_METAL || kind == MAT_ISOTROPIC;
}

template <uint32_t F, bool COUNT>
RT_HD bool material_scatter_with_sample(const SceneView& sv, const FlatMaterial& m, const Ray& r_in,
                                        const HitRecord& rec, Rng& g, Vec3 sphere_sample, Ray* scattered,
                                        Color* attenuation, TraceCounters* cnt) {
  if (COUNT) cnt->scatters++;
  // Lambertian normalises its sphere sample (random_unit_vector, vec3.rs:297-299), Metal and Dielectric normalise the incoming
  // direction (hit.rs:1070, 1109): one `v / v.length()` -- a square root and three divisions in f64 -- per material, on different
  // operands.  Written inside the branches a wave executes it once per material kind present, each time for that kind's lanes
  // only; hoisted in front of the switch it runs ONCE for all lanes on each lane's own operand.  Same function of the same
  // operand: the same bits.  (Lanes of kinds that normalise nothing compute a value nobody reads.)
  const Vec3 unit_in = unit(((F & F_LAMBERTIAN) && m.kind == MAT_LAMBERTIAN) ? sphere_sample : r_in.direction);
  if ((F & F_LAMBERTIAN) && m.kind == MAT_LAMBERTIAN) {  // hit.rs:1039-1051
    Vec3 scatter_direction = rec.normal + unit_in;  // random_unit_vector, vec3.rs:297-299
    if (near_zero(scatter_direction)) scatter_direction = rec.normal;
    *scattered = make_ray(rec.p, scatter_direction, r_in.time);
    *attenuation = material_texture_value<F, COUNT>(sv, m, rec, cnt);
    return true;
  }
  if ((F & F_METAL) && m.kind == MAT_METAL) {  // hit.rs:1069-1083 (fuzz sphere drawn even when fuzz == 0)
    Vec3 reflected = reflect(unit_in, rec.normal);
    Vec3 dir = reflected + m.param * sphere_sample;
    *scattered = make_ray(rec.p, dir, r_in.time);
    *attenuation = load_v3(m.albedo);
    return dot(dir, rec.normal) > real(0.0);
  }
  if ((F & F_DIELECTRIC) && m.kind == MAT_DIELECTRIC) {  // hit.rs:1103-1126 (uniform drawn only if refraction is possible)
    // 1 / ir and reflectance's r0^2 = ((1 - x) / (1 + x))^2 for x = 1 / ir (front face) and x = ir (back face) depend on the
    // material alone: the flattener evaluates exactly these expressions once (albedo[0..2] of a Dielectric record, flatten.cpp)
    // instead of two divisions per glass hit -- correctly rounded IEEE divisions on both sides, the same bits.
    real refraction_ratio = rec.front_face ? m.albedo[0] : m.param;
    Vec3 unit_direction = unit_in;
    real cos_theta = rt_fmin(dot(-unit_direction, rec.normal), real(1.0));
    real sin_theta = rt_sqrt(real(1.0) - cos_theta * cos_theta);
    bool cannot_refract = refraction_ratio * sin_theta > real(1.0);
    Vec3 direction;
    if (cannot_refract || reflectance_r0sq(cos_theta, rec.front_face ? m.albedo[1] : m.albedo[2]) > rng_f64(g))
      direction = reflect(unit_direction, rec.normal);
    else
      direction = refract(unit_direction, rec.normal, refraction_ratio);
    *scattered = make_ray(rec.p, direction, r_in.time);
    *attenuation = v3(1, 1, 1);
    return true;
  }
  if ((F & F_ISOTROPIC) && m.kind == MAT_ISOTROPIC) {  // hit.rs:1005-1010
    *scattered = make_ray(rec.p, sphere_sample, r_in.time);
    *attenuation = material_texture_value<F, COUNT>(sv, m, rec, cnt);
    return true;
  }
  return false;  // MAT_DIFFUSE_LIGHT, hit.rs:1146-1148
}

template <uint32_t F, bool COUNT>
RT_HD bool material_scatter(const SceneView& sv, const FlatMaterial& m, const Ray& r_in,
                            const HitRecord& rec, Rng& g, Ray* scattered, Color* attenuation,
                            TraceCounters* cnt) {
  Vec3 sphere_sample = v3(0, 0, 0);
  if (material_needs_sphere_sample(m.kind)) sphere_sample = random_in_unit_sphere(g);
  return material_scatter_with_sample<F, COUNT>(sv, m, r_in, rec, g, sphere_sample, scattered, attenuation, cnt);
}

}  // namespace rt
