// Deterministic many-value reductions for 64-wide wavefronts (gfx950).
#pragma once
#include <hip/hip_runtime.h>

namespace gfs_red {

// Wave-wide sums of N <= 32 per-lane values by recursive halving: at level s the lanes of a pair (lane ^ 2^s) split the
// remaining values between them (one keeps the even, the other the odd ones, each adds what its partner sends), so the work
// per level halves: 16 + 8 + 4 + 2 + 1 exchanges instead of N x 6 for one shuffle tree per value.  On return lane l (and lane
// l + 32) holds the wave total of value (l & 31).  Fixed summation order.
template <int N>
__device__ __forceinline__ double wave_sum_many(const double (&vals)[N]) {
  static_assert(N <= 32, "at most 32 values");
  const int lane = threadIdx.x & 63;
  double v[32];
#pragma unroll
  for (int k = 0; k < 32; k++) v[k] = k < N ? vals[k] : 0.0;
#pragma unroll
  for (int s = 0; s < 5; s++) {
    const bool odd = (lane >> s) & 1;
    const int half = 16 >> s;
#pragma unroll
    for (int j = 0; j < half; j++) {
      const double keep = odd ? v[2 * j + 1] : v[2 * j];
      const double send = odd ? v[2 * j] : v[2 * j + 1];
      v[j] = keep + __shfl_xor(send, 1 << s, 64);
    }
  }
  return v[0] + __shfl_xor(v[0], 32, 64);
}

// Sums of N <= 32 values over a workgroup of WAVES wavefronts: out[k] (k < N) is valid in threads 0 .. N-1 after the call.
// s_buf: WAVES * 32 doubles of LDS.  Two barriers in total (not two per value).
template <int N, int WAVES>
__device__ __forceinline__ double block_sum_many(const double (&vals)[N], double* s_buf) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const double tot = wave_sum_many<N>(vals);
  __syncthreads();  // s_buf may still be read from a previous call
  if (lane < 32) s_buf[wave * 32 + lane] = tot;
  __syncthreads();
  double r = 0;
  if (threadIdx.x < 32) {
#pragma unroll
    for (int w = 0; w < WAVES; w++) r += s_buf[w * 32 + threadIdx.x];
  }
  return r;
}

}  // namespace gfs_red
