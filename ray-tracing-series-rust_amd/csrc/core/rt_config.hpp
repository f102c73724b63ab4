// Shared host/device arithmetic core -- build configuration.
//
// Every header under core/ is compiled twice: by hipcc for gfx950 (the product's
// kernels) and by g++ for the host (scene construction in the product, and the
// CPU checkers under oracle/).  Both compilations MUST use -ffp-contract=off:
// the reference is Rust, which never contracts a*b+c into an FMA, and the parity
// gate between the HIP path and the CPU oracle is bit-exact f64.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define RT_HD __host__ __device__ __forceinline__
#define RT_HD_NOINLINE __host__ __device__ __attribute__((noinline)) inline
#else
#define RT_HD inline
#define RT_HD_NOINLINE __attribute__((noinline)) inline
#endif

// Keep a loop rolled (the device compiler otherwise unrolls Perlin's 7 octaves x 8 corners and keeps
// every gradient fetch live: hundreds of registers for a rarely taken path).
#if defined(__clang__)
#define RT_NO_UNROLL _Pragma("clang loop unroll(disable)")
#elif defined(__GNUC__)
#define RT_NO_UNROLL _Pragma("GCC unroll 1")
#else
#define RT_NO_UNROLL
#endif

// Load a record whose address is the same in every lane of a wave (the scene's top-level table).
// On the device the read goes through the constant address space, which lets the compiler use
// scalar loads (one fetch per wave into SGPRs) instead of 64 identical vector loads.
#if defined(__HIP_DEVICE_COMPILE__)
template <class T>
__device__ __forceinline__ T rt_load_uniform(const T* p) {
  return *(const __attribute__((address_space(4))) T*)(p);
}
#else
template <class T>
inline T rt_load_uniform(const T* p) { return *p; }
#endif

// true if the predicate holds for any lane the wave is executing (device) / for this thread (host)
#if defined(__HIP_DEVICE_COMPILE__)
#define RT_WAVE_ANY(p) (__builtin_amdgcn_ballot_w64(p) != 0ull)
#else
#define RT_WAVE_ANY(p) (p)
#endif

#if defined(__HIP_DEVICE_COMPILE__)
#define RT_DEVICE_CODE 1
#else
#define RT_DEVICE_CODE 0
#endif

// The arithmetic type of the path.  double = the parity build (everything above: bit-identical to the CPU oracle);
// float = the fast mode, compiled as a second translation unit into namespace rt32 (hip/render_f32.hip):
// same source, `real` = float, judged statistically, never against the oracle's bits.
#ifndef RT_REAL
#define RT_REAL double
#endif

namespace rt {

typedef RT_REAL real;

// Bit casts (memcpy-free so they stay in registers on the device).
RT_HD uint64_t f64_bits(double x) {
  union { double d; uint64_t u; } c; c.d = x; return c.u;
}
RT_HD double bits_f64(uint64_t u) {
  union { double d; uint64_t u; } c; c.u = u; return c.d;
}
RT_HD uint32_t f64_hi(double x) { return (uint32_t)(f64_bits(x) >> 32); }
RT_HD uint32_t f64_lo(double x) { return (uint32_t)(f64_bits(x)); }
RT_HD double f64_from_words(uint32_t hi, uint32_t lo) {
  return bits_f64(((uint64_t)hi << 32) | (uint64_t)lo);
}

#define RT_INFINITY ((::rt::real)__builtin_huge_val())

// IEEE-exact primitives used by the core.  sqrt/fabs/floor are correctly
// rounded (or exact) on both x86-64 and gfx950, so they are safe to share.
RT_HD double rt_sqrt(double x) { return __builtin_sqrt(x); }
RT_HD double rt_fabs(double x) { return __builtin_fabs(x); }
RT_HD double rt_floor(double x) { return __builtin_floor(x); }
// Rust f64::min / f64::max: a NaN operand is ignored (== C fmin/fmax).
RT_HD double rt_fmin(double a, double b) { return __builtin_fmin(a, b); }
RT_HD double rt_fmax(double a, double b) { return __builtin_fmax(a, b); }
RT_HD bool rt_isnan(double x) { return x != x; }
// the same primitives for real = float
RT_HD float rt_sqrt(float x) { return __builtin_sqrtf(x); }
RT_HD float rt_fabs(float x) { return __builtin_fabsf(x); }
RT_HD float rt_floor(float x) { return __builtin_floorf(x); }
RT_HD float rt_fmin(float a, float b) { return __builtin_fminf(a, b); }
RT_HD float rt_fmax(float a, float b) { return __builtin_fmaxf(a, b); }
RT_HD bool rt_isnan(float x) { return x != x; }

// Rust `x as i32` for f64: saturating, NaN -> 0 (vec3.rs:103-105, perlin.rs:33-35,
// texture.rs:107-108).
RT_HD int32_t rt_f64_as_i32(double x) {
  if (rt_isnan(x)) return 0;
  if (x >= 2147483647.0) return 2147483647;
  if (x <= -2147483648.0) return (int32_t)(-2147483647 - 1);
  return (int32_t)x;
}
RT_HD int32_t rt_f64_as_i32(float x) { return rt_f64_as_i32((double)x); }

}  // namespace rt
