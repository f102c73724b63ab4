// Transcendental functions built only from IEEE +,-,*,/ and sqrt.
//
// The reference calls f64::sin (texture.rs:57,86), f64::ln (hit.rs:969),
// f64::acos / f64::atan2 (hit.rs:197-198), f64::tan (camera.rs:27) and
// f64::sin/cos (hit.rs:845-846) from the platform libm.  glibc (host) and OCML
// (gfx950) differ in the last bit on these, and two of the call sites gate
// branches (the checker sign test and the medium's free-path test), which would
// make a bit-exact GPU-vs-CPU parity gate impossible.  So the core carries its
// own implementations: classic argument-reduction + minimax-polynomial kernels
// (the algorithms published with Sun's fdlibm), expressed with nothing but
// correctly rounded basic operations.  Compiled with -ffp-contract=off they
// return bit-identical results on x86-64 and gfx950; tests/test_rt_math.py
// checks them against numpy's libm to <= 2 ulp.
#pragma once
#include "rt_config.hpp"

namespace rt {

#define RT_PI ((::rt::real)3.14159265358979323846)  // std::f64::consts::PI

namespace detail {

// sin on [-pi/4, pi/4], y = tail of x.
RT_HD double k_sin(double x, double y, int iy) {
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
               S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
               S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  double z = x * x;
  double w = z * z;
  double r = S2 + z * (S3 + z * S4) + z * w * (S5 + z * S6);
  double v = z * x;
  if (iy == 0) return x + v * (S1 + z * r);
  return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}

// cos on [-pi/4, pi/4], y = tail of x.
RT_HD double k_cos(double x, double y) {
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
               C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
               C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  double z = x * x;
  double w = z * z;
  double r = z * (C1 + z * (C2 + z * C3)) + (w * w) * (C4 + z * (C5 + z * C6));
  double hz = 0.5 * z;
  w = 1.0 - hz;
  return w + (((1.0 - w) - hz) + (z * r - x * y));
}

// x = n*(pi/2) + (y0 + y1), |y0| <= pi/4.  Two-step Cody-Waite reduction with a
// 33+33+53-bit split of pi/2; accurate for |x| < 2^20 * pi/2 (~1.6e6), far
// beyond 10*p or scale*p.z for any scene coordinate the reference uses (<= 5000).
RT_HD int rem_pio2(double x, double* y0, double* y1) {
  const double invpio2 = 6.36619772367581382433e-01;
  const double pio2_1 = 1.57079632673412561417e+00, pio2_1t = 6.07710050650619224932e-11;
  const double pio2_2 = 6.07710050630396597660e-11, pio2_2t = 2.02226624879595063154e-21;
  double fn = (x * invpio2 + 0x1.8p52) - 0x1.8p52;  // round to nearest integer
  double r = x - fn * pio2_1;                        // exact: pio2_1 has 33 bits
  double t = r;
  double w = fn * pio2_2;
  r = t - w;
  w = fn * pio2_2t - ((t - r) - w);
  (void)pio2_1t;
  double a = r - w;
  *y0 = a;
  *y1 = (r - a) - w;
  return (int)(int64_t)fn;
}

}  // namespace detail

RT_HD double rt_sin(double x) {
  if (!(rt_fabs(x) <= 1.0e300)) return x - x;  // inf/NaN -> NaN
  if (rt_fabs(x) <= 0.78539816339744830962) {
    if (rt_fabs(x) < 0x1.0p-26) return x;
    return detail::k_sin(x, 0.0, 0);
  }
  double y0, y1;
  int n = detail::rem_pio2(x, &y0, &y1);
  switch (n & 3) {
    case 0: return detail::k_sin(y0, y1, 1);
    case 1: return detail::k_cos(y0, y1);
    case 2: return -detail::k_sin(y0, y1, 1);
    default: return -detail::k_cos(y0, y1);
  }
}

RT_HD double rt_cos(double x) {
  if (!(rt_fabs(x) <= 1.0e300)) return x - x;
  if (rt_fabs(x) <= 0.78539816339744830962) {
    if (rt_fabs(x) < 0x1.0p-27) return 1.0;
    return detail::k_cos(x, 0.0);
  }
  double y0, y1;
  int n = detail::rem_pio2(x, &y0, &y1);
  switch (n & 3) {
    case 0: return detail::k_cos(y0, y1);
    case 1: return -detail::k_sin(y0, y1, 1);
    case 2: return -detail::k_cos(y0, y1);
    default: return detail::k_sin(y0, y1, 1);
  }
}

// The SIGN of rt_sin(x) without evaluating the polynomials: -1, 0, +1, or 2 when rt_sin(x) is NaN.  Exact, not an
// approximation: on the reduced range k_cos is positive (>= 0.7), and k_sin(y0, y1) = y0 - (terms smaller than |y0|),
// so it has the sign of y0 (of the tail y1 in the unreachable case y0 == 0); the quadrant does the rest.  The checker
// texture (texture.rs:56-62) only asks whether sin(10x) sin(10y) sin(10z) < 0 -- a product of three factors no smaller
// than ~1e-19 each cannot underflow, so its sign is the product of these signs.
RT_HD int rt_sin_sign(double x) {
  if (!(rt_fabs(x) <= 1.0e300)) return (x - x == 0.0) ? 0 : 2;  // rt_sin returns x - x there: 0 if finite, else NaN
  if (rt_fabs(x) <= 0.78539816339744830962) return x > 0.0 ? 1 : (x < 0.0 ? -1 : 0);
  double y0, y1;
  int n = detail::rem_pio2(x, &y0, &y1);
  int s = y0 > 0.0 ? 1 : (y0 < 0.0 ? -1 : (y1 > 0.0 ? 1 : (y1 < 0.0 ? -1 : 0)));
  switch (n & 3) {
    case 0: return s;
    case 1: return 1;
    case 2: return -s;
    default: return -1;
  }
}

// Host-only use (Camera::new); quotient of the two kernels above.
RT_HD double rt_tan(double x) { return rt_sin(x) / rt_cos(x); }

// Natural logarithm: x = 2^k (1+f), s = f/(2+f), log(1+f) = f - f^2/2 + s (f^2/2 + R(s^2)).
RT_HD double rt_log(double x) {
  const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
  const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
               Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
               Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
               Lg7 = 1.479819860511658591e-01;
  int32_t hx = (int32_t)f64_hi(x);
  uint32_t lx = f64_lo(x);
  int32_t k = 0;
  if (hx < 0x00100000) {  // x < 2^-1022, zero or negative
    if (((hx & 0x7fffffff) | (int32_t)lx) == 0) return -RT_INFINITY;  // log(+-0) = -inf
    if (hx < 0) return (x - x) / (x - x);                               // log(-#) = NaN
    k -= 54;
    x *= 0x1.0p54;  // scale up a subnormal
    hx = (int32_t)f64_hi(x);
    lx = f64_lo(x);
  }
  if (hx >= 0x7ff00000) return x + x;  // inf or NaN
  k += (hx >> 20) - 1023;
  hx &= 0x000fffff;
  int32_t i = (hx + 0x95f64) & 0x100000;
  x = f64_from_words((uint32_t)(hx | (i ^ 0x3ff00000)), lx);  // normalize x or x/2
  k += (i >> 20);
  double f = x - 1.0;
  double s = f / (2.0 + f);
  double dk = (double)k;
  double z = s * s;
  double w = z * z;
  double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
  double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
  double R = t2 + t1;
  double hfsq = 0.5 * f * f;
  return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
}

RT_HD double rt_acos(double x) {
  const double pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17;
  const double pi = 3.14159265358979311600e+00;
  const double pS0 = 1.66666666666666657415e-01, pS1 = -3.25565818622400915405e-01,
               pS2 = 2.01212532134862925881e-01, pS3 = -4.00555345006794114027e-02,
               pS4 = 7.91534994289814532176e-04, pS5 = 3.47933107596021167570e-05;
  const double qS1 = -2.40339491173441421878e+00, qS2 = 2.02094576023350569471e+00,
               qS3 = -6.88283971605453293030e-01, qS4 = 7.70381505559019352791e-02;
  double ax = rt_fabs(x);
  if (!(ax < 1.0)) {
    if (x == 1.0) return 0.0;
    if (x == -1.0) return pi + 2.0 * pio2_lo;
    return (x - x) / (x - x);  // |x| > 1 or NaN
  }
  if (ax < 0.5) {
    if (ax <= 0x1.0p-57) return pio2_hi + pio2_lo;
    double z = x * x;
    double p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    double q = 1.0 + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    double r = p / q;
    return pio2_hi - (x - (pio2_lo - x * r));
  }
  if (x < 0.0) {  // x < -0.5
    double z = (1.0 + x) * 0.5;
    double p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    double q = 1.0 + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    double s = rt_sqrt(z);
    double r = p / q;
    double w = r * s - pio2_lo;
    return pi - 2.0 * (s + w);
  }
  // x > 0.5
  double z = (1.0 - x) * 0.5;
  double s = rt_sqrt(z);
  double df = f64_from_words(f64_hi(s), 0u);
  double c = (z - df * df) / (s + df);
  double p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
  double q = 1.0 + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
  double r = p / q;
  double w = r * s + c;
  return 2.0 * (df + w);
}

RT_HD double rt_atan(double x) {
  const double aT0 = 3.33333333333329318027e-01, aT1 = -1.99999999998764832476e-01,
               aT2 = 1.42857142725034663711e-01, aT3 = -1.11111104054623557880e-01,
               aT4 = 9.09088713343650656196e-02, aT5 = -7.69187620504482999495e-02,
               aT6 = 6.66107313738753120669e-02, aT7 = -5.83357013379057348645e-02,
               aT8 = 4.97687799461593236017e-02, aT9 = -3.65315727442169155270e-02,
               aT10 = 1.62858201153657823623e-02;
  if (rt_isnan(x)) return x + x;
  double ax = rt_fabs(x);
  bool neg = (f64_hi(x) >> 31) != 0;
  if (ax >= 0x1.0p66) {
    double r = 1.57079632679489655800e+00 + 6.12323399573676603587e-17;
    return neg ? -r : r;
  }
  int id;
  double hi = 0.0, lo = 0.0;
  if (ax < 0.4375) {
    if (ax < 0x1.0p-29) return x;
    id = -1;
  } else {
    x = ax;
    if (ax < 1.1875) {
      if (ax < 0.6875) {
        id = 0; x = (2.0 * x - 1.0) / (2.0 + x);
        hi = 4.63647609000806093515e-01; lo = 2.26987774529616870924e-17;
      } else {
        id = 1; x = (x - 1.0) / (x + 1.0);
        hi = 7.85398163397448278999e-01; lo = 3.06161699786838301793e-17;
      }
    } else {
      if (ax < 2.4375) {
        id = 2; x = (x - 1.5) / (1.0 + 1.5 * x);
        hi = 9.82793723247329054082e-01; lo = 1.39033110312309984516e-17;
      } else {
        id = 3; x = -1.0 / x;
        hi = 1.57079632679489655800e+00; lo = 6.12323399573676603587e-17;
      }
    }
  }
  double z = x * x;
  double w = z * z;
  double s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
  double s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
  if (id < 0) return x - x * (s1 + s2);
  z = hi - ((x * (s1 + s2) - lo) - x);
  return neg ? -z : z;
}

RT_HD double rt_atan2(double y, double x) {
  const double pi = 3.1415926535897931160E+00, pi_lo = 1.2246467991473531772E-16;
  const double pi_o_2 = 1.5707963267948965580E+00, pi_o_4 = 7.8539816339744827900E-01;
  if (rt_isnan(x) || rt_isnan(y)) return x + y;
  uint32_t hx = f64_hi(x), hy = f64_hi(y);
  int m = (int)((hy >> 31) & 1u) | (int)((hx >> 30) & 2u);  // 2*sign(x) + sign(y)
  if (y == 0.0) {
    switch (m) {
      case 0: case 1: return y;        // atan(+-0, +anything) = +-0
      case 2: return pi;               // atan(+0, -anything) = pi
      default: return -pi;             // atan(-0, -anything) = -pi
    }
  }
  if (x == 0.0) return (hy >> 31) ? -pi_o_2 : pi_o_2;
  double ax = rt_fabs(x), ay = rt_fabs(y);
  if (ax == RT_INFINITY) {
    if (ay == RT_INFINITY) {
      switch (m) {
        case 0: return pi_o_4;
        case 1: return -pi_o_4;
        case 2: return 3.0 * pi_o_4;
        default: return -3.0 * pi_o_4;
      }
    }
    switch (m) {
      case 0: return 0.0;
      case 1: return -0.0;
      case 2: return pi;
      default: return -pi;
    }
  }
  if (ay == RT_INFINITY) return (hy >> 31) ? -pi_o_2 : pi_o_2;
  int32_t k = ((int32_t)(hy & 0x7fffffffu) - (int32_t)(hx & 0x7fffffffu)) >> 20;
  double z;
  if (k > 60) z = pi_o_2 + 0.5 * pi_lo;          // |y/x| > 2^60
  else if ((hx >> 31) && k < -60) z = 0.0;        // |y|/x < -2^-60
  else z = rt_atan(rt_fabs(y / x));
  switch (m) {
    case 0: return z;
    case 1: return -z;
    case 2: return pi - (z - pi_lo);
    default: return (z - pi_lo) - pi;
  }
}

// f64::to_radians (camera.rs:26, hit.rs:844): x * (PI / 180)
RT_HD double rt_to_radians(double deg) { return deg * (3.14159265358979323846 / 180.0); }

#if defined(RT_F32)
// real = float (the fast mode, hip/render_f32.hip, which includes <cmath>): the platform's single-precision functions;
// no parity claim there.
RT_HD float rt_sin(float x) { return ::sinf(x); }
RT_HD float rt_cos(float x) { return ::cosf(x); }
RT_HD float rt_log(float x) { return ::logf(x); }
RT_HD float rt_acos(float x) { return ::acosf(x); }
RT_HD float rt_atan2(float y, float x) { return ::atan2f(y, x); }
RT_HD int rt_sin_sign(float x) {
  const float s = ::sinf(x);
  return s != s ? 2 : (s > 0.0f ? 1 : (s < 0.0f ? -1 : 0));
}
#endif

}  // namespace rt
