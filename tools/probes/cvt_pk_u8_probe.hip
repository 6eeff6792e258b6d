// Probe: does v_cvt_pk_u8_f32 round like saturate_cast<uchar>(cvRound(x)) (round-half-even, clamp to 0..255)?
// hipcc --offload-arch=gfx950 -O2 tools/probes/cvt_pk_u8_probe.hip -o /tmp/cvt_probe && /tmp/cvt_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
__global__ void k(const float* x, unsigned* a, unsigned* b, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  a[i] = __builtin_amdgcn_cvt_pk_u8_f32(x[i], 0u, 0u);
  const int r = __float2int_rn(x[i]);
  b[i] = (unsigned)min(max(r, 0), 255);
}
int main() {
  const int n = 1 << 24;
  float* hx = new float[n];
  // every float between -4 and 260 at steps of 1/64, their neighbours one ulp either side, and a sweep of raw bit patterns
  int m = 0;
  for (int q = -4 * 64; q <= 260 * 64 && m + 3 < n; q++) {
    const float v = q / 64.0f;
    unsigned u;
    memcpy(&u, &v, 4);
    hx[m++] = v;
    unsigned lo = u - 1, hi = u + 1;
    memcpy(&hx[m++], &lo, 4);
    memcpy(&hx[m++], &hi, 4);
  }
  for (unsigned u = 0x3f000000u; m < n; u += 37) memcpy(&hx[m++], &u, 4);  // 0.5 .. large
  float* dx;
  unsigned *da, *db;
  hipMalloc(&dx, n * 4);
  hipMalloc(&da, n * 4);
  hipMalloc(&db, n * 4);
  hipMemcpy(dx, hx, n * 4, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(dx, da, db, n);
  unsigned* ha = new unsigned[n];
  unsigned* hb = new unsigned[n];
  hipMemcpy(ha, da, n * 4, hipMemcpyDeviceToHost);
  hipMemcpy(hb, db, n * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < n; i++)
    if (ha[i] != hb[i] && hx[i] == hx[i]) {
      if (bad < 10) printf("x=%.9g cvt_pk=%u ref=%u\n", hx[i], ha[i], hb[i]);
      bad++;
    }
  printf("CVTPROBE n=%d mismatches=%d\n", n, bad);
  return 0;
}
