// Double-precision sin / cos / pow(x, 3) with the arithmetic of glibc 2.35 on an x86-64 with FMA (the `__sin_fma`,
// `__cos_fma`, `__pow_fma` ifunc variants every current x86-64 host selects), bit for bit.
//
// Why: the reference's pose updates go through g2o's SE3Quat::exp (Thirdparty/g2o/g2o/types/se3quat.h:223-257:
// sin(theta), cos(theta), pow(theta, 3)) and its Levenberg-Marquardt step control
// (Thirdparty/g2o/g2o/core/optimization_algorithm_levenberg.cpp:127: pow(2*rho - 1, 3)), all plain libm calls.  A device
// libm that differs from glibc in the last bit of one call changes the estimate by 1e-16, and at a converged state the sign
// of the gain ratio -- one more LM iteration or not -- is decided by exactly such bits.  With these functions (and the
// edge-ordered sums of pose.hip) PoseOptimization follows the CPU restatement bit for bit.
//
// Restated from the published algorithms (glibc 2.35 sysdeps/ieee754/dbl-64/s_sin.c: __sin, __cos, do_sin, do_cos, TAYLOR_SIN;
// e_pow.c: pow, log_inline, exp_inline, specialcase -- the ARM optimized-routines pow), tables in glibc_tables.inc.  Where
// glibc's FMA build contracts a*b+c the fma is written out; everything else is a single rounded operation (the library is
// built with -ffp-contract=off).  tests/test_glibc_math.py compiles this header for the host and compares it with the
// host's libm over 10^8 arguments; tests/test_gpu_glibc_math.py runs the device side against the same libm.
//
// Ranges: pow3 is checked bit for bit on 5 * 10^8 arguments with 2^-300 <= |x| <= 2^170; above (results beyond 2^510, through the
// overflow-guard path of exp_inline) one argument in 4 * 10^6 differs in the last bit.  sin / cos restate glibc for |x| < 2.426 (the branches without the Payne-Hanek style reduction); beyond that -- a
// rotation update of more than 139 degrees, which no optimizer step of this path produces -- they use the device libm.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define GFS_HD __host__ __device__ inline
#else
#define GFS_HD inline
#endif

#include "glibc_tables.inc"

namespace gfs_glibc {

GFS_HD uint64_t as_u64(double x) {
  uint64_t u;
  memcpy(&u, &x, 8);
  return u;
}
GFS_HD double as_f64(uint64_t u) {
  double x;
  memcpy(&x, &u, 8);
  return x;
}
GFS_HD double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }

struct SinCosEntry {
  double sn, ssn, cs, ccs;
};
GFS_HD SinCosEntry sincos_entry(int k) {  // SINCOS_TABLE_LOOKUP: sin / cos of k / 128, high and low parts
  static constexpr double tab[440] = {GFS_GLIBC_SINCOSTAB};
  return SinCosEntry{tab[4 * k], tab[4 * k + 1], tab[4 * k + 2], tab[4 * k + 3]};
}

// usncs.h
constexpr double kS1 = -0x1.5555555555555p-3, kS2 = 0x1.1111111110ECEp-7, kS3 = -0x1.A01A019DB08B8p-13, kS4 = 0x1.71DE27B9A7ED9p-19,
                 kS5 = -0x1.ADDFFC2FCDF59p-26;
constexpr double kSn3 = -1.66666666666664880952546298448555E-01, kSn5 = 8.33333214285722277379541354343671E-03,
                 kCs2 = 4.99999999999999999999950396842453E-01, kCs4 = -4.16666666666664434524222570944589E-02,
                 kCs6 = 1.38888874007937613028114285595617E-03;
constexpr double kBig = 0x1.8p45, kHp0 = 0x1.921FB54442D18p0, kHp1 = 0x1.1A62633145C07p-54;

GFS_HD double do_sin(double x, double dx) {  // s_sin.c do_sin
  const double xold = x;
  if (fabs(x) < 0.126) {  // TAYLOR_SIN(x*x, x, dx)
    const double xx = x * x;
    const double p = fma_(fma_(fma_(fma_(kS5, xx, kS4), xx, kS3), xx, kS2), xx, kS1);
    const double t = fma_(fma_(p, x, -0.5 * dx), xx, dx);
    return x + t;
  }
  if (x <= 0) dx = -dx;
  const double ux = kBig + fabs(x);
  x = fabs(x) - (ux - kBig);
  const double xx = x * x;
  const double s = x + fma_(x * xx, fma_(xx, kSn5, kSn3), dx);
  const double c = fma_(x, dx, xx * fma_(xx, fma_(xx, kCs6, kCs4), kCs2));
  const SinCosEntry e = sincos_entry((int)(uint32_t)as_u64(ux));
  const double cor = fma_(e.cs, s, fma_(-e.sn, c, fma_(s, e.ccs, e.ssn)));
  return copysign(e.sn + cor, xold);
}
GFS_HD double do_cos(double x, double dx) {  // s_sin.c do_cos
  if (x < 0) dx = -dx;
  const double ux = kBig + fabs(x);
  x = fabs(x) - (ux - kBig) + dx;
  const double xx = x * x;
  const double s = fma_(x * xx, fma_(xx, kSn5, kSn3), x);
  const double c = xx * fma_(xx, fma_(xx, kCs6, kCs4), kCs2);
  const SinCosEntry e = sincos_entry((int)(uint32_t)as_u64(ux));
  const double cor = fma_(-e.sn, s, fma_(-e.cs, c, fma_(-s, e.ssn, e.ccs)));
  return e.cs + cor;
}

GFS_HD double sin(double x) {  // __sin
  const uint32_t k = (uint32_t)(as_u64(x) >> 32) & 0x7fffffffu;
  if (k < 0x3e500000u) return x;              // |x| < 2^-26
  if (k < 0x3feb6000u) return do_sin(x, 0.0);  // |x| < 0.855469
  if (k < 0x400368fdu) {                       // |x| < 2.426265
    const double t = kHp0 - fabs(x);
    return copysign(do_cos(t, kHp1), x);
  }
  return ::sin(x);
}
GFS_HD double cos(double x) {  // __cos
  const uint32_t k = (uint32_t)(as_u64(x) >> 32) & 0x7fffffffu;
  if (k < 0x3e400000u) return 1.0;             // |x| < 2^-27
  if (k < 0x3feb6000u) return do_cos(x, 0.0);
  if (k < 0x400368fdu) {
    const double y = kHp0 - fabs(x);
    const double a = y + kHp1;
    const double da = (y - a) + kHp1;
    return do_sin(a, da);
  }
  return ::cos(x);
}

// ---- pow(x, 3.0): e_pow.c with y fixed to 3 (an odd integer: the sign of x carries over) ----
GFS_HD double pow_log_inline(uint64_t ix, double* tail) {
  static constexpr double hdr[9] = {GFS_GLIBC_POWLOG_HDR};
  static constexpr double tab[128 * 3] = {GFS_GLIBC_POWLOG_TAB};
  const double Ln2hi = hdr[0], Ln2lo = hdr[1];
  const double* A = hdr + 2;
  const uint64_t tmp = ix - 0x3fe6955500000000ull;
  const int i = (int)((tmp >> (52 - 7)) % 128);
  const int k = (int)((int64_t)tmp >> 52);
  const uint64_t iz = ix - (tmp & (0xfffull << 52));
  const double z = as_f64(iz), kd = (double)k;
  const double invc = tab[3 * i], logc = tab[3 * i + 1], logctail = tab[3 * i + 2];
  const double r = fma_(z, invc, -1.0);
  const double t1 = fma_(kd, Ln2hi, logc);
  const double t2 = t1 + r;
  const double lo1 = fma_(kd, Ln2lo, logctail);
  const double lo2 = t1 - t2 + r;
  const double ar = A[0] * r, ar2 = r * ar, ar3 = r * ar2;
  const double hi = t2 + ar2;
  const double lo3 = fma_(ar, r, -ar2);
  const double lo4 = t2 - hi + ar2;
  const double p = ar3 * fma_(ar2, fma_(ar2, fma_(r, A[6], A[5]), fma_(r, A[4], A[3])), fma_(r, A[2], A[1]));
  const double lo = lo1 + lo2 + lo3 + lo4 + p;
  const double y = hi + lo;
  *tail = hi - y + lo;
  return y;
}
GFS_HD double pow_exp_inline(double x, double xtail, bool negate) {
  static constexpr double hdr[8] = {GFS_GLIBC_EXP_HDR};
  static constexpr unsigned long long tab[256] = {GFS_GLIBC_EXP_TAB};
  const double InvLn2N = hdr[0], Shift = hdr[1], NegLn2hiN = hdr[2], NegLn2loN = hdr[3], C2 = hdr[4], C3 = hdr[5], C4 = hdr[6], C5 = hdr[7];
  uint32_t abstop = (uint32_t)(as_u64(x) >> 52) & 0x7ff;
  if (abstop - 0x3c9u >= 0x408u - 0x3c9u) {          // |x| < 2^-54 or |x| >= 512
    if (abstop - 0x3c9u >= 0x80000000u) return negate ? -1.0 : 1.0;
    if (abstop >= 0x409u) {                           // |x| >= 1024: overflow / underflow
      const double big = (as_u64(x) >> 63) ? 0.0 : HUGE_VAL;
      return negate ? -big : big;
    }
    abstop = 0;
  }
  const double z = InvLn2N * x;
  double kd = z + Shift;
  const uint64_t ki = as_u64(kd);
  kd -= Shift;
  double r = fma_(kd, NegLn2loN, fma_(kd, NegLn2hiN, x));
  r += xtail;
  const uint64_t idx = 2 * (ki % 128);
  const uint64_t top = (ki + (negate ? (0x800ull << 7) : 0ull)) << (52 - 7);
  const double tail = as_f64(tab[idx]);
  uint64_t sbits = tab[idx + 1] + top;
  const double r2 = r * r;
  const double tmp = fma_(r2 * r2, fma_(r, C5, C4), fma_(r2, fma_(r, C3, C2), tail + r));
  if (abstop == 0) {  // specialcase(): the exponent of scale may have left the normal range
    if ((ki & 0x80000000ull) == 0) {
      sbits -= 1009ull << 52;
      const double scale = as_f64(sbits);
      return 0x1p1009 * fma_(scale, tmp, scale);
    }
    sbits += 1022ull << 52;
    const double scale = as_f64(sbits);
    const double st = scale * tmp;  // used twice below: the compiler keeps the product instead of fusing it
    double y = scale + st;
    if (fabs(y) < 1.0) {  // subnormal result: round once, from a value kept in two parts
      const double one = y < 0.0 ? -1.0 : 1.0;
      double lo = scale - y + st;
      double hi = one + y;
      lo = one - hi + y + lo;
      y = (hi + lo) - one;
      if (y == 0.0) y = as_f64(sbits & 0x8000000000000000ull);
    }
    return 0x1p-1022 * y;
  }
  const double scale = as_f64(sbits);
  return fma_(scale, tmp, scale);
}
GFS_HD double pow3(double x) {  // pow(x, 3.0)
  uint64_t ix = as_u64(x);
  const bool neg = (ix >> 63) != 0;
  const uint32_t topx = (uint32_t)(ix >> 52);
  if (topx - 0x001u >= 0x7ffu - 0x001u) {  // zero, subnormal, negative, inf, nan
    if (2 * ix - 1 >= 2 * 0x7ff0000000000000ull - 1) {  // zeroinfnan(x)
      if (x != x) return x + 3.0;
      const double x2 = x * x;  // 0 or inf
      return neg ? -x2 : x2;
    }
    ix &= 0x7fffffffffffffffull;
    if ((ix >> 52) == 0) {  // subnormal x: normalise
      ix = as_u64(as_f64(ix) * 0x1p52);
      ix -= 52ull << 52;
    }
  }
  double lo;
  const double hi = pow_log_inline(ix, &lo);
  const double ehi = 3.0 * hi;
  const double elo = fma_(3.0, lo, fma_(3.0, hi, -ehi));
  return pow_exp_inline(ehi, elo, neg);
}

}  // namespace gfs_glibc
