// Shared host-side infrastructure of libgfs_hip.so: error reporting, HIP checks, device buffers,
// profiled kernel launches.  gfx950 (MI355X) only.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/gfs_abi.h"
#include "../../include/gfs_abi_test.h"  // the test hooks are compiled into the library; their prototypes are not part of the boundary

namespace gfs {

void set_error(const char* fmt, ...);
bool device_ok(int device);  // true iff `device` exists and is a gfx950 part

#define GFS_HIP(call)                                                                              \
  do {                                                                                             \
    hipError_t _e = (call);                                                                        \
    if (_e != hipSuccess) {                                                                        \
      ::gfs::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, __LINE__); \
      return GFS_ERR_HIP;                                                                          \
    }                                                                                              \
  } while (0)

#define GFS_REQUIRE(cond, code, ...)  \
  do {                                \
    if (!(cond)) {                    \
      ::gfs::set_error(__VA_ARGS__);  \
      return code;                    \
    }                                 \
  } while (0)

// RAII device / pinned-host buffers (sized once at handle creation: nothing is allocated on the hot path).
template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  int alloc(size_t count) {
    free();
    n = count;
    if (count == 0) return GFS_OK;
    GFS_HIP(hipMalloc((void**)&p, count * sizeof(T)));
    return GFS_OK;
  }
  void free() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
  ~DevBuf() { free(); }
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
};

template <typename T>
struct PinBuf {
  T* p = nullptr;
  size_t n = 0;
  int alloc(size_t count) {
    free();
    n = count;
    if (count == 0) return GFS_OK;
    GFS_HIP(hipHostMalloc((void**)&p, count * sizeof(T), hipHostMallocDefault));
    return GFS_OK;
  }
  void free() {
    if (p) (void)hipHostFree(p);
    p = nullptr;
    n = 0;
  }
  ~PinBuf() { free(); }
  PinBuf() = default;
  PinBuf(const PinBuf&) = delete;
  PinBuf& operator=(const PinBuf&) = delete;
};

// ---- optional per-kernel timing (gfs_profile_*): HIP events recorded on the launch stream ----
bool profile_on();
int profile_begin(const char* name, hipStream_t s);  // returns a record id for profile_end (thread-safe)
void profile_end(int id, hipStream_t s);

#define GFS_LAUNCH(name, kernel, grid, block, shmem, stream, ...)                        \
  do {                                                                                   \
    const int _pid = ::gfs::profile_on() ? ::gfs::profile_begin(name, stream) : -1;      \
    hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);                 \
    if (_pid >= 0) ::gfs::profile_end(_pid, stream);                                     \
    hipError_t _le = hipGetLastError();                                                  \
    if (_le != hipSuccess) {                                                             \
      ::gfs::set_error("launch %s failed: %s (%s:%d)", name, hipGetErrorString(_le), __FILE__, __LINE__); \
      return GFS_ERR_HIP;                                                                \
    }                                                                                    \
  } while (0)

inline int div_up(int a, int b) { return (a + b - 1) / b; }
#if defined(__HIPCC__)
// min / max over the 64 lanes of a wave (all lanes must be active); every lane gets the result.  Used in front of LDS atomics that a
// whole workgroup would otherwise aim at one address (they serialise: ~35 cycles each).
__device__ __forceinline__ int wave_min_i32(int v) {
  for (int o = 32; o >= 1; o >>= 1) v = min(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ int wave_max_i32(int v) {
  for (int o = 32; o >= 1; o >>= 1) v = max(v, __shfl_xor(v, o));
  return v;
}
// Walks p[i0], p[i0 + stride], ... (index < n) with U loads in flight and hands every element to f(index, value) in index order.
// A plain grid-stride loop over global memory is compiled into load - wait - use per trip: with the 17-odd trips a thread of a
// one-workgroup-per-cloud kernel makes, that is 17 dependent round trips of ~1.5 us where two or three would do.
// (P: any pointer type — an address-space-qualified one keeps its address space; decltype(+p[0]) is the plain element type)
template <int U, class P, class F>
__device__ __forceinline__ void strided_batch(P p, int i0, int stride, int n, F&& f) {
  using T = decltype(+p[0]);
  for (int i = i0; i < n; i += U * stride) {
    T t[U];
#pragma unroll
    for (int u = 0; u < U; u++) t[u] = p[min(i + u * stride, n - 1)];  // (clamped: no branch in front of a load)
#pragma unroll
    for (int u = 0; u < U; u++)
      if (i + u * stride < n) f(i + u * stride, t[u]);
  }
}
#endif

inline size_t align_up(size_t a, size_t b) { return (a + b - 1) / b * b; }

}  // namespace gfs
