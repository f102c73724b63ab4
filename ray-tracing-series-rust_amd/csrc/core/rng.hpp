// Counter-based per-(pixel, sample) random streams.
//
// The reference draws every random number from rand::thread_rng() (OS-seeded
// ChaCha12; call sites: /root/reference/src/world.rs:1212-1213, camera.rs:61,69,
// vec3.rs:273-322, hit.rs:969,1007,1040,1074,1118), so it has no reproducible
// stream to match.  This build replaces it with a stream that is a pure function
// of (render seed, global pixel index, sample index):
//
//   stream state  = Philox4x32-10( counter = {pixel_lo, pixel_hi, sample, 0},
//                                  key     = {seed_lo, seed_hi} )     (128 bits)
//   k-th draw     = k-th output of xoroshiro128+ started from that state (only its high 53 / 52 bits are ever used)
//   gen::<f64>()      = (u64 >> 11) * 2^-53  in [0,1)   (53 high bits: rand 0.8's Standard
//                                                        distribution for f64)
//   gen_range(a..b)   = v12 * (b - a) + (a - (b - a)),  v12 = bits(0x3FF0.. | u64 >> 12) in [1,2)
//                       (52 mantissa bits: rand 0.8's UniformFloat::sample_single, without its
//                        redraw when rounding lands exactly on b)
//
// Philox (Salmon et al., SC'11) gives statistically independent streams for
// every (pixel, sample) at one evaluation per path; xoroshiro128+ (Blackman &
// Vigna; the variant its authors recommend for floating-point generation from the upper
// bits -- its weak LOW bits are shifted out here) is multiply-free, which matters on CDNA
// where 32-bit integer multiplies are quarter rate and a path draws ~20-30 doubles, and it
// is one 64-bit add and one rotation cheaper per draw than xoroshiro128++ (rounds 1 and 2 up
// to here: 46 instead of 58 issue clocks per draw in the rejection loop that every
// Lambertian / Metal / Isotropic scatter runs).  Scheduling (which lane,
// wave, GPU or CPU thread runs a sample) cannot change any draw.
#pragma once
#include "rt_config.hpp"

namespace rt {

// a ^ b ^ c: ONE instruction on gfx950 (v_bitop3_b32 with the truth table 0x96), which the compiler does not pick by itself.
RT_HD uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96u);
#else
  return a ^ b ^ c;
#endif
}


RT_HD void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
  const uint32_t W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)M0 * (uint64_t)c[0];
    uint64_t p1 = (uint64_t)M1 * (uint64_t)c[2];
    uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    uint32_t n0 = xor3(hi1, c[1], k0);
    uint32_t n1 = lo1;
    uint32_t n2 = xor3(hi0, c[3], k1);
    uint32_t n3 = lo0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += W0; k1 += W1;
  }
}

struct Rng {
  uint64_t s0, s1;
};

// 0 < k < 64.  On the device a 64-bit rotate is two v_alignbit_b32 (the generic shift/shift/or form costs four
// instructions there); same value either way.
RT_HD uint64_t rotl64(uint64_t x, int k) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  u32x2 v = __builtin_bit_cast(u32x2, x);  // v.x = low half
  if (k >= 32) { v = v.yx; k -= 32; }
  if (k == 0) return __builtin_bit_cast(uint64_t, v);
  u32x2 r;
  r.y = __builtin_amdgcn_alignbit(v.y, v.x, (uint32_t)(32 - k));  // (hi:lo) >> (32-k) = hi << k | lo >> (32-k)
  r.x = __builtin_amdgcn_alignbit(v.x, v.y, (uint32_t)(32 - k));
  return __builtin_bit_cast(uint64_t, r);
#else
  return (x << k) | (x >> (64 - k));
#endif
}

RT_HD Rng rng_for_sample(uint64_t seed, uint64_t pixel_index, uint32_t sample) {
  uint32_t c[4];
  c[0] = (uint32_t)pixel_index;
  c[1] = (uint32_t)(pixel_index >> 32);
  c[2] = sample;
  c[3] = 0u;
  philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  Rng r;
  r.s0 = (uint64_t)c[0] | ((uint64_t)c[1] << 32);
  r.s1 = (uint64_t)c[2] | ((uint64_t)c[3] << 32);
  if ((r.s0 | r.s1) == 0) r.s0 = 0x9E3779B97F4A7C15ull;  // xoroshiro must not start at 0
  return r;
}

// xoroshiro128+ 1.0 (a=24, b=16, c=37; output s0 + s1).
RT_HD uint64_t rng_next_u64(Rng& r) {
  uint64_t s0 = r.s0, s1 = r.s1;
  uint64_t result = s0 + s1;
  s1 ^= s0;
#if defined(__HIP_DEVICE_COMPILE__)
  // xor3 of the 32-bit halves: two instructions fewer per draw, six per attempt of the rejection loops (the hottest loop of every
  // workload)
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  const u32x2 a = __builtin_bit_cast(u32x2, rotl64(s0, 24)), b = __builtin_bit_cast(u32x2, s1), c = __builtin_bit_cast(u32x2, s1 << 16);
  u32x2 n;
  n.x = xor3(a.x, b.x, c.x);
  n.y = xor3(a.y, b.y, c.y);
  r.s0 = __builtin_bit_cast(uint64_t, n);
#else
  r.s0 = rotl64(s0, 24) ^ s1 ^ (s1 << 16);
#endif
  r.s1 = rotl64(s1, 37);
  return result;
}

RT_HD double u64_to_unit_f64(uint64_t x) {
  // (x >> 11) has at most 53 significant bits: the conversion and the scaling
  // by a power of two are both exact.
  return (double)(x >> 11) * 0x1.0p-53;
}

// rng.gen::<f64>()  (real = float: the 24 high bits of the same 64-bit output, times 2^-24)
#if defined(RT_F32)
RT_HD real rng_f64(Rng& r) { return (float)(uint32_t)(rng_next_u64(r) >> 40) * 0x1.0p-24f; }
#else
RT_HD double rng_f64(Rng& r) { return u64_to_unit_f64(rng_next_u64(r)); }
#endif
// rng.gen_range(lo..hi) for f64: the [1,2) mantissa construction of rand 0.8.5's
// UniformFloat<f64>::sample_single -- value1_2 * scale + (low - scale).
RT_HD double u64_to_f64_1_2(uint64_t x) { return bits_f64((x >> 12) | 0x3FF0000000000000ull); }
#if defined(RT_F32)
RT_HD real rng_range(Rng& r, real lo, real hi) {  // the same [1,2) construction with the 23 mantissa bits of a float
  union { uint32_t u; float f; } c;
  c.u = (uint32_t)(rng_next_u64(r) >> 41) | 0x3F800000u;
  const float scale = hi - lo, offset = lo - scale;
  return c.f * scale + offset;
}
#else
RT_HD double rng_range(Rng& r, double lo, double hi) {
  double scale = hi - lo;
  double offset = lo - scale;
  return u64_to_f64_1_2(rng_next_u64(r)) * scale + offset;
}
#endif

// rng.gen_range(-1.0..1.0), the draw of the rejection samplers (vec3.rs:288-294, 310-322): value1_2 * 2 + (-3) in rand 0.8.5's
// construction.  value1_2 * 2 is the same mantissa one exponent up, and 2 v - 3 is a multiple of 2^-51 below 1 in magnitude:
// every step is exact, so building the number in [2, 4) directly and adding -3 returns the same bits with one f64
// instruction less per draw (three per attempt of the rejection loop, the hottest loop of every workload).
#if defined(RT_F32)
RT_HD real rng_range_pm1(Rng& r) {
  union { uint32_t u; float f; } c;
  c.u = (uint32_t)(rng_next_u64(r) >> 41) | 0x40000000u;  // [2, 4)
  return c.f + -3.0f;
}
#else
RT_HD double rng_range_pm1(Rng& r) { return bits_f64((rng_next_u64(r) >> 12) | 0x4000000000000000ull) + -3.0; }
#endif

// SplitMix64: host-side generator for scene construction (random sphere
// placement, box heights, Perlin tables, the oracle's reference-rule BVH axis).
struct HostRng {
  uint64_t state;
};
RT_HD uint64_t host_rng_next_u64(HostRng& h) {
  uint64_t z = (h.state += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
RT_HD double host_rng_f64(HostRng& h) { return u64_to_unit_f64(host_rng_next_u64(h)); }
RT_HD double host_rng_range(HostRng& h, double lo, double hi) {
  double scale = hi - lo;
  double offset = lo - scale;
  return u64_to_f64_1_2(host_rng_next_u64(h)) * scale + offset;
}
// gen_range(lo..hi) for integers, lo < hi.
RT_HD uint64_t host_rng_below(HostRng& h, uint64_t n) { return host_rng_next_u64(h) % n; }

}  // namespace rt
