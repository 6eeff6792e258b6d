// Host-side check of geoflowslam_amd/csrc/glibc_math.hpp against this machine's libm (glibc 2.35, FMA variants).
// usage: glibc_math_check N  -> prints "sin <bad> cos <bad> pow3 <bad> of <N>"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../geoflowslam_amd/csrc/glibc_math.hpp"

static bool same(double a, double b) { return (a != a && b != b) || memcmp(&a, &b, 8) == 0; }

int main(int argc, char** argv) {
  const long N = argc > 1 ? atol(argv[1]) : 1000000;
  uint64_t st = 88172645463325252ull;
  long bad_s = 0, bad_c = 0, bad_p = 0;
  volatile double three = 3.0;
  auto check = [&](double x) {
    volatile double vx = x;
    if (!same(std::sin(vx), gfs_glibc::sin(x))) { if (bad_s++ < 5) printf("sin %a\n", x); }
    if (!same(std::cos(vx), gfs_glibc::cos(x))) { if (bad_c++ < 5) printf("cos %a\n", x); }
    if (!same(std::pow(vx, three), gfs_glibc::pow3(x))) { if (bad_p++ < 5) printf("pow3 %a: %a vs %a\n", x, std::pow(vx, three), gfs_glibc::pow3(x)); }
  };
  const double special[] = {0.0, -0.0, 1.0, -1.0, 0.126, 0.125, 0.855469, 0.8554687, 2.426265, 2.4262, 1e-5, 1e-8, 7.450580596923828e-09, 3.725290298461914e-09,
                            1e300, -1e300, 1e-300, -1e-300, 1e-110, 5e-324, -5e-324, 2.2250738585072014e-308, 1e103, INFINITY, -INFINITY, NAN,
                            0.5, 2.0, 3.0, 1.0000000000000002, 0.9999999999999999};
  for (double x : special) check(x);
  for (long i = 0; i < N; i++) {
    st ^= st << 13; st ^= st >> 7; st ^= st << 17;
    double x = ((st >> 11) * (1.0 / 9007199254740992.0) * 2.0 - 1.0) * 3.2;
    const int mode = i & 3;
    if (mode == 1) x = ldexp(x, -(int)((st >> 3) & 31));
    if (mode == 2) x = ldexp(x, (int)((st >> 3) % 470) - 300);  // pow3 through its large-|y log x| paths (results 2^-900 .. 2^510)
    check(x);
  }
  printf("sin %ld cos %ld pow3 %ld of %ld\n", bad_s, bad_c, bad_p, N);
  return (bad_s || bad_c || bad_p) ? 1 : 0;
}
