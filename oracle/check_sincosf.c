/* TEST INFRASTRUCTURE (oracle side) — not product code.
 *
 * Exhaustive check that the explicit restatement of glibc's single-precision cosf/sinf
 * (sysdeps/ieee754/flt-32/s_sincosf.h, glibc >= 2.28; x86-64 FMA ifunc variant) used by the
 * HIP rBRIEF kernel is bit-identical to this host's libm for every float in [0, 6.5].
 * The reference calls cosf/sinf at src/ORBextractor.cc:101-102 with angle in [0, 2*pi].
 *
 *   gcc -O2 -mfma -o check_sincosf check_sincosf.c -lm -fopenmp && ./check_sincosf
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

static const double C0 = 0x1p0, C1 = -0x1.ffffffd0c621cp-2, C2 = 0x1.55553e1068f19p-5,
                    C3 = -0x1.6c087e89a359dp-10, C4 = 0x1.99343027bf8c3p-16;
static const double S1 = -0x1.555545995a603p-3, S2 = 0x1.1107605230bc4p-7, S3 = -0x1.994eb3774cf24p-13;
static const double HPI_INV = 0x1.45F306DC9C883p+23, HPI = 0x1.921FB54442D18p0;

static inline uint32_t asuint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline uint32_t abstop12(float x) { return (asuint(x) >> 20) & 0x7ff; }

#ifdef NOFMA
#define FMA(a, b, c) ((a) * (b) + (c))
#else
#define FMA(a, b, c) fma((a), (b), (c))
#endif

/* flip = 1 negates the polynomial (table entry [1] of glibc) */
static inline float poly(double x, double x2, int n, int flip) {
  if ((n & 1) == 0) {
    double x3 = x * x2;
    double s1 = FMA(x2, S3, S2);
    double x7 = x3 * x2;
    double s = FMA(x3, S1, x);
    return (float)FMA(x7, s1, s);
  } else {
    double sg = flip ? -1.0 : 1.0;
    double x4 = x2 * x2;
    double c2 = FMA(x2, sg * C4, sg * C3);
    double c1 = FMA(x2, sg * C1, sg * C0);
    double x6 = x4 * x2;
    double c = FMA(x4, sg * C2, c1);
    return (float)FMA(x6, c2, c);
  }
}

static inline double reduce_fast(double x, int* np) {
  double r = x * HPI_INV;
  int n = ((int32_t)r + 0x800000) >> 24;
  *np = n;
  return FMA(-(double)n, HPI, x);
}

static const double SIGN[4] = {1.0, -1.0, -1.0, 1.0};

float gfs_sinf(float y) {
  double x = y;
  if (abstop12(y) < abstop12(0x1.921FB6p-1f)) {
    if (abstop12(y) < abstop12(0x1p-12f)) return y;
    return poly(x, x * x, 0, 0);
  }
  int n;
  x = reduce_fast(x, &n);
  double s = SIGN[n & 3];
  return poly(x * s, x * x, n, (n & 2) != 0);
}

float gfs_cosf(float y) {
  double x = y;
  if (abstop12(y) < abstop12(0x1.921FB6p-1f)) {
    if (abstop12(y) < abstop12(0x1p-12f)) return 1.0f;
    return poly(x, x * x, 1, 0);
  }
  int n;
  x = reduce_fast(x, &n);
  double s = SIGN[n & 3];
  return poly(x * s, x * x, n ^ 1, (n & 2) != 0);
}

int main(void) {
  const uint32_t hi = asuint(6.5f);
  long bad_s = 0, bad_c = 0;
#pragma omp parallel for reduction(+ : bad_s, bad_c) schedule(static)
  for (uint32_t u = 0; u <= hi; ++u) {
    float f;
    memcpy(&f, &u, 4);
    float a = sinf(f), b = gfs_sinf(f);
    float c = cosf(f), d = gfs_cosf(f);
    if (asuint(a) != asuint(b)) bad_s++;
    if (asuint(c) != asuint(d)) bad_c++;
  }
  printf("checked %u floats in [0,6.5]: sinf mismatches=%ld cosf mismatches=%ld\n", hi + 1, bad_s, bad_c);
  return (bad_s || bad_c) ? 1 : 0;
}
