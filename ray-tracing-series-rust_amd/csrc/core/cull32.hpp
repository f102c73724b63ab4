// Conservative single-precision box culling for BVH walks.
//
// The reference decides every hit with f64 primitive tests (hit.rs) and uses Aabb::hit
// (aabb.rs:23-61) only to skip subtrees.  Skipping is an optimisation: if a box test never says
// "miss" for a box the ray really enters, the set of primitive tests that can win is unchanged
// and so is the closest hit (with its tie rule).  This header provides such a test in f32:
//
//   * child boxes are rounded outward to f32 on the host (lo down, hi up);
//   * the ray is rounded to f32 once per bounce: o32 = fl(o), i32 ~ 1/d, oi = fl(o32*i32).  On the device
//     i32 = v_rcp_f32(fl(d)) (1 ulp; an f64 division here cost 45 VALU per bounce), on the host fl(1/d): with
//     u = 2^-24, i32 = (1/d)(1 + e1), |e1| <= 3u in the worse (device) case;
//   * per axis t = fma(plane, i32, -oi); the slab interval is [min, max] of the two planes;
//   * first-order error of each t against the real-arithmetic value T = (plane - o)/d:
//         t - T = T e1 - (o/d)(e2 + e3) + t e4     (e2: rounding of o, e3: of the product oi, e4: of the fma)
//         |t - T| <= 4u |t| + 2u |oi|
//     so the interval is widened by |t| * 2^-21 + E with E = 2^-21 * max_axis |oi| (8u |t| + 8u |oi|: twice and
//     four times the bound, which also swallows the second-order terms and the rounding of the widening
//     itself).  Because f(t) = t - |t| eps is monotone the widening is applied once, after the max / min over axes,
//     and as ONE comparison: "widened near > widened far"  <=>  tn - tf > (|tn| + |tf|) 2^-21 + 2E
//     (4 instructions per box instead of 6; the rounding of the difference, <= u (|tn| + |tf|), sits inside the slack);
//   * t_min is rounded down, the running closest distance is rounded up;
//   * a box passes unless the widened interval is provably empty (NaNs pass).
// A zero direction component gives i32 = inf and NaN/inf slab values; fminf/fmaxf ignore NaN so
// that axis simply stops culling (still conservative).  Host and device need not agree bit for bit
// here -- any conservative answer gives the same final hit -- so FMA use is fine.
#pragma once
#include "flat_types.hpp"

namespace rt {

struct Ray32 {
  float ix, iy, iz;     // fl(1/d)
  float oix, oiy, oiz;  // fl(o32 * i32)
  float err2;           // 2E, E = 2^-21 * max |oi| over axes with finite slope
  float t_min;          // rounded down
};

RT_HD float cull_round_up(real x) {  // an f32 >= x (inf stays inf)
  return (float)x * (x >= 0.0 ? 1.00000012f : 0.99999988f);
}

RT_HD Ray32 make_ray32(const Ray& r, real t_min) {
  Ray32 q;
  float ox = (float)r.origin.x, oy = (float)r.origin.y, oz = (float)r.origin.z;
#if defined(__HIP_DEVICE_COMPILE__)
  q.ix = __builtin_amdgcn_rcpf((float)r.direction.x);
  q.iy = __builtin_amdgcn_rcpf((float)r.direction.y);
  q.iz = __builtin_amdgcn_rcpf((float)r.direction.z);
#else
  q.ix = (float)(1.0 / r.direction.x);
  q.iy = (float)(1.0 / r.direction.y);
  q.iz = (float)(1.0 / r.direction.z);
#endif
#if defined(RT_F32)
  // Single-precision rays do have direction components of exactly 0 (a cancellation leaves nothing below one ulp), a few in
  // every 10^8 rays.  With 1/d = inf the planes of that axis come out as +-inf or NaN, the widening below becomes inf and no
  // box can be ruled out any more: such a ray walked the WHOLE tree, alone in its wave (the dragon room: 4 rays, 0.4 s).
  // The fast mode gives the axis a finite slope of 2^-60 instead: a ray outside the slab sees both planes at the same huge
  // distance and is culled, a ray inside sees (-huge, +huge).  The axis is left out of the error term like an infinite one
  // (its distances only matter through their sign).
  const float slope_cap = 0x1.0p60f;
#if defined(__HIP_DEVICE_COMPILE__)
  q.ix = __builtin_amdgcn_fmed3f(q.ix, -slope_cap, slope_cap);
  q.iy = __builtin_amdgcn_fmed3f(q.iy, -slope_cap, slope_cap);
  q.iz = __builtin_amdgcn_fmed3f(q.iz, -slope_cap, slope_cap);
#else
  q.ix = __builtin_fminf(__builtin_fmaxf(q.ix, -slope_cap), slope_cap);
  q.iy = __builtin_fminf(__builtin_fmaxf(q.iy, -slope_cap), slope_cap);
  q.iz = __builtin_fminf(__builtin_fmaxf(q.iz, -slope_cap), slope_cap);
#endif
  const float finite_slope = 0x1.0p60f;
#else
  const float finite_slope = 1e30f;
#endif
  q.oix = ox * q.ix; q.oiy = oy * q.iy; q.oiz = oz * q.iz;
  float ax = __builtin_fabsf(q.ix) < finite_slope ? __builtin_fabsf(q.oix) : 0.0f;
  float ay = __builtin_fabsf(q.iy) < finite_slope ? __builtin_fabsf(q.oiy) : 0.0f;
  float az = __builtin_fabsf(q.iz) < finite_slope ? __builtin_fabsf(q.oiz) : 0.0f;
  q.err2 = __builtin_fmaxf(ax, __builtin_fmaxf(ay, az)) * 0x1.0p-20f;
  // rounded DOWN whatever the sign (a medium's second boundary query may start at a negative t)
  q.t_min = (float)t_min * (t_min >= 0.0 ? 0.99999988f : 1.00000012f);
  return q;
}

// true unless the ray provably misses the box within (t_min, t_max32]
RT_HD bool cull32_may_hit(const float* lo, const float* hi, const Ray32& q, float t_max32) {
  float ax = __builtin_fmaf(lo[0], q.ix, -q.oix), bx = __builtin_fmaf(hi[0], q.ix, -q.oix);
  float ay = __builtin_fmaf(lo[1], q.iy, -q.oiy), by = __builtin_fmaf(hi[1], q.iy, -q.oiy);
  float az = __builtin_fmaf(lo[2], q.iz, -q.oiz), bz = __builtin_fmaf(hi[2], q.iz, -q.oiz);
  float tn = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(ax, bx), __builtin_fminf(ay, by)),
                             __builtin_fmaxf(__builtin_fminf(az, bz), q.t_min));
  float tf = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(ax, bx), __builtin_fmaxf(ay, by)),
                             __builtin_fminf(__builtin_fmaxf(az, bz), t_max32));
  return !(tn - tf > __builtin_fmaf(__builtin_fabsf(tn) + __builtin_fabsf(tf), 0x1.0p-21f, q.err2));
}

// The same test with the near / far plane of every axis already picked by the sign of the ray's direction (near = lo
// where the ray travels towards +axis, hi otherwise).  For a finite slope the near value IS min(a, b) and the far value
// max(a, b) of cull32_may_hit, so tn / tf -- and the verdict -- are the same numbers; for a zero direction component
// both are NaN and fmaxf / fminf drop them, exactly as there.  What it saves is the six min / max per box that only
// sorted the two planes: on gfx950 v_min_f32 / v_max_f32 issue in 4 clocks, twice a v_fma_f32.
RT_HD bool cull32_may_hit_nf(float nx, float fx, float ny, float fy, float nz, float fz, const Ray32& q, float t_max32) {
  const float tn = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(__builtin_fmaf(nx, q.ix, -q.oix), __builtin_fmaf(ny, q.iy, -q.oiy)),
                                                   __builtin_fmaf(nz, q.iz, -q.oiz)), q.t_min);
  const float tf = __builtin_fminf(__builtin_fminf(__builtin_fminf(__builtin_fmaf(fx, q.ix, -q.oix), __builtin_fmaf(fy, q.iy, -q.oiy)),
                                                   __builtin_fmaf(fz, q.iz, -q.oiz)), t_max32);
  return !(tn - tf > __builtin_fmaf(__builtin_fabsf(tn) + __builtin_fabsf(tf), 0x1.0p-21f, q.err2));
}

// The same verdict for queries whose t_min is POSITIVE (every ray_color query: world.hit(r, 0.001, inf), world.rs:77 -- not the
// second boundary query of a medium), written so that it costs three instructions after the min / max: with tn >= t_min > 0,
//     tn - tf > (tn + |tf|) e + 2E   <=>   tn (1 - e) - 2E  >  tf + |tf| e            (e = 2^-21; 1 - e is an exact f32)
// one fma per side (|tf| is a free source modifier) and one compare; the roundings of the two fmas are of the size of the one
// the difference had (<= u (tn + |tf|)), inside the same slack.
RT_HD bool cull32_may_hit_nf_pos(float nx, float fx, float ny, float fy, float nz, float fz, const Ray32& q, float t_max32) {
  const float tn = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(__builtin_fmaf(nx, q.ix, -q.oix), __builtin_fmaf(ny, q.iy, -q.oiy)),
                                                   __builtin_fmaf(nz, q.iz, -q.oiz)), q.t_min);
  const float tf = __builtin_fminf(__builtin_fminf(__builtin_fminf(__builtin_fmaf(fx, q.ix, -q.oix), __builtin_fmaf(fy, q.iy, -q.oiy)),
                                                   __builtin_fmaf(fz, q.iz, -q.oiz)), t_max32);
  return !(__builtin_fmaf(tn, 1.0f - 0x1.0p-21f, -q.err2) > __builtin_fmaf(__builtin_fabsf(tf), 0x1.0p-21f, tf));
}

// Both children of a node at once.  Same arithmetic as two cull32_may_hit calls; on the device the twelve plane
// fmas are written as four 2-wide ones (x and y of a plane set) plus four scalar ones for z, so that the pairs
// issue as v_pk_fma_f32 -- an fma gives the same value whichever instruction carries it.
RT_HD void cull32_may_hit2(const float* lo0, const float* hi0, const float* lo1, const float* hi1, const Ray32& q,
                           float t_max32, bool* hit0, bool* hit1) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef float f2 __attribute__((ext_vector_type(2)));
  const f2 ixy = {q.ix, q.iy}, oxy = {-q.oix, -q.oiy};
  const float noz = -q.oiz;
  // x and y of one plane set are adjacent in memory (they arrive as one register pair): pack those
  const f2 a0 = __builtin_elementwise_fma((f2){lo0[0], lo0[1]}, ixy, oxy), b0 = __builtin_elementwise_fma((f2){hi0[0], hi0[1]}, ixy, oxy);
  const f2 a1 = __builtin_elementwise_fma((f2){lo1[0], lo1[1]}, ixy, oxy), b1 = __builtin_elementwise_fma((f2){hi1[0], hi1[1]}, ixy, oxy);
  const float az0 = __builtin_fmaf(lo0[2], q.iz, noz), bz0 = __builtin_fmaf(hi0[2], q.iz, noz);
  const float az1 = __builtin_fmaf(lo1[2], q.iz, noz), bz1 = __builtin_fmaf(hi1[2], q.iz, noz);
  float tn0 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(a0.x, b0.x), __builtin_fminf(a0.y, b0.y)),
                              __builtin_fmaxf(__builtin_fminf(az0, bz0), q.t_min));
  float tf0 = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(a0.x, b0.x), __builtin_fmaxf(a0.y, b0.y)),
                              __builtin_fminf(__builtin_fmaxf(az0, bz0), t_max32));
  float tn1 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(a1.x, b1.x), __builtin_fminf(a1.y, b1.y)),
                              __builtin_fmaxf(__builtin_fminf(az1, bz1), q.t_min));
  float tf1 = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(a1.x, b1.x), __builtin_fmaxf(a1.y, b1.y)),
                              __builtin_fminf(__builtin_fmaxf(az1, bz1), t_max32));
  *hit0 = !(tn0 - tf0 > __builtin_fmaf(__builtin_fabsf(tn0) + __builtin_fabsf(tf0), 0x1.0p-21f, q.err2));
  *hit1 = !(tn1 - tf1 > __builtin_fmaf(__builtin_fabsf(tn1) + __builtin_fabsf(tf1), 0x1.0p-21f, q.err2));
#else
  *hit0 = cull32_may_hit(lo0, hi0, q, t_max32);
  *hit1 = cull32_may_hit(lo1, hi1, q, t_max32);
#endif
}

}  // namespace rt
