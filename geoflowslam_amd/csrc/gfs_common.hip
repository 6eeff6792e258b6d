// Error reporting, device probing, timers and the per-kernel profiler of libgfs_hip.so.
#include "gfs_common.hpp"

#include <algorithm>
#include <map>

namespace gfs {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

bool device_ok(int device) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) {
    (void)hipGetLastError();
    set_error("no usable HIP device %d (count=%d): libgfs_hip has no CPU fallback", device, n);
    return false;
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) {
    set_error("hipGetDeviceProperties(%d) failed", device);
    return false;
  }
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    set_error("device %d is %s; libgfs_hip is built for gfx950 (MI355X) only", device, prop.gcnArchName);
    return false;
  }
  return true;
}

// ---------------- profiler ----------------
struct ProfRec {
  std::string name;
  hipEvent_t e0, e1;
};
static bool g_prof = false;
static std::mutex g_prof_mu;
static std::vector<ProfRec> g_pending;
static std::vector<hipEvent_t> g_pool;
static std::map<std::string, std::pair<double, int64_t>> g_acc;

bool profile_on() { return g_prof; }

static hipEvent_t get_event() {
  if (!g_pool.empty()) {
    hipEvent_t e = g_pool.back();
    g_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  (void)hipEventCreate(&e);
  return e;
}

int profile_begin(const char* name, hipStream_t s) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  ProfRec r;
  r.name = name;
  r.e0 = get_event();
  r.e1 = get_event();
  (void)hipEventRecord(r.e0, s);
  g_pending.push_back(r);
  return (int)g_pending.size() - 1;
}

void profile_end(int id, hipStream_t s) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (id >= 0 && id < (int)g_pending.size()) (void)hipEventRecord(g_pending[id].e1, s);
}

static void profile_drain() {
  for (ProfRec& r : g_pending) {
    (void)hipEventSynchronize(r.e1);
    float ms = 0;
    if (hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) {
      auto& a = g_acc[r.name];
      a.first += ms;
      a.second += 1;
    }
    g_pool.push_back(r.e0);
    g_pool.push_back(r.e1);
  }
  g_pending.clear();
}

}  // namespace gfs

struct gfs_timer {
  int device;
  hipEvent_t e0, e1;
};

namespace {
__global__ void k_cal_stream_read(const double4* __restrict__ a, long long n, double* __restrict__ sink) {
  double acc = 0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const double4 v = a[i];
    acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 12345.678) sink[threadIdx.x] = acc;  // never true: keeps the loads
}
__global__ void k_cal_gather32(const double4* __restrict__ a, long long n, long long table, int per_thread, double* __restrict__ sink) {
  double acc = 0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    unsigned long long h = (unsigned long long)i * 0x9E3779B97F4A7C15ull + 0x7F4A7C15ull;
    for (int k = 0; k < per_thread; k++) {
      h ^= h >> 29;
      h *= 0xBF58476D1CE4E5B9ull;
      const double4 v = a[(long long)(h % (unsigned long long)table)];
      acc += v.x + v.y + v.z + v.w;
    }
  }
  if (acc == 12345.678) sink[threadIdx.x] = acc;
}
__global__ void k_cal_stream_write(double4* __restrict__ a, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    a[i] = make_double4((double)i, 1.0, 2.0, 3.0);
}
}  // namespace

extern "C" {

int gfs_abi_version(void) { return GFS_ABI_VERSION; }
const char* gfs_last_error(void) { return gfs::g_err; }

int gfs_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  int ok = 0;
  for (int d = 0; d < n; d++) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, d) == hipSuccess && strncmp(prop.gcnArchName, "gfx950", 6) == 0) ok++;
  }
  return ok;
}

int gfs_timer_create(int device, gfs_timer** out) {
  GFS_REQUIRE(out, GFS_ERR_INVALID_ARG, "gfs_timer_create: out is NULL");
  if (!gfs::device_ok(device)) return GFS_ERR_NO_DEVICE;
  GFS_HIP(hipSetDevice(device));
  gfs_timer* t = new gfs_timer;
  t->device = device;
  GFS_HIP(hipEventCreate(&t->e0));
  GFS_HIP(hipEventCreate(&t->e1));
  *out = t;
  return GFS_OK;
}
void gfs_timer_destroy(gfs_timer* t) {
  if (!t) return;
  (void)hipEventDestroy(t->e0);
  (void)hipEventDestroy(t->e1);
  delete t;
}
int gfs_timer_start(gfs_timer* t, void* stream) {
  GFS_HIP(hipEventRecord(t->e0, (hipStream_t)stream));
  return GFS_OK;
}
int gfs_timer_stop(gfs_timer* t, void* stream) {
  GFS_HIP(hipEventRecord(t->e1, (hipStream_t)stream));
  return GFS_OK;
}
int gfs_timer_elapsed_ms(gfs_timer* t, float* ms) {
  GFS_HIP(hipEventSynchronize(t->e1));
  GFS_HIP(hipEventElapsedTime(ms, t->e0, t->e1));
  return GFS_OK;
}

int gfs_profile_enable(int on) {
  std::lock_guard<std::mutex> lk(gfs::g_prof_mu);
  gfs::g_prof = on != 0;
  return GFS_OK;
}
int gfs_profile_reset(void) {
  std::lock_guard<std::mutex> lk(gfs::g_prof_mu);
  gfs::profile_drain();
  gfs::g_acc.clear();
  return GFS_OK;
}
int gfs_profile_report(char (*names)[64], double* total_ms, int64_t* launches, int cap) {
  std::lock_guard<std::mutex> lk(gfs::g_prof_mu);
  gfs::profile_drain();
  int i = 0;
  for (auto& kv : gfs::g_acc) {
    if (i < cap) {
      snprintf(names[i], 64, "%s", kv.first.c_str());
      total_ms[i] = kv.second.first;
      launches[i] = kv.second.second;
    }
    i++;
  }
  return i;
}

// Test hook for calibrating the HBM counters (profiles/calibrate.sh): kernels with KNOWN byte counts in the access patterns of the
// hot kernels.  mode 0: streaming read of n 32-byte records (k_cal_stream_read); mode 1: every thread gathers `per_thread` 32-byte
// records at pseudo-random indices of a table of `table` records (k_cal_gather32: the per-lane point gathers of the k-NN / 1-NN
// searches); mode 2: streaming write of n 32-byte records (k_cal_stream_write).  Returns the bytes the kernel asked for.
int gfs_test_traffic(int device, int mode, long long n, long long table, int per_thread, long long* bytes_out) {
  GFS_REQUIRE(bytes_out && n > 0 && mode >= 0 && mode <= 2, GFS_ERR_INVALID_ARG, "gfs_test_traffic: invalid argument");
  if (!gfs::device_ok(device)) return GFS_ERR_NO_DEVICE;
  GFS_HIP(hipSetDevice(device));
  gfs::DevBuf<double4> d_a;
  gfs::DevBuf<double> d_sink;
  const long long alloc = mode == 1 ? table : n;
  int rc = d_a.alloc((size_t)alloc);
  if (!rc) rc = d_sink.alloc(1 << 20);
  if (rc) return rc;
  GFS_HIP(hipMemset(d_a.p, 0, (size_t)alloc * sizeof(double4)));
  GFS_HIP(hipMemset(d_sink.p, 0, (size_t)(1 << 20) * sizeof(double)));
  GFS_HIP(hipDeviceSynchronize());
  const int blocks = (int)std::min<long long>((n + 255) / 256, 1 << 20);
  if (mode == 0) {
    GFS_LAUNCH("k_cal_stream_read", k_cal_stream_read, dim3(blocks), dim3(256), 0, (hipStream_t)0, d_a.p, n, d_sink.p);
    *bytes_out = n * 32;
  } else if (mode == 1) {
    GFS_LAUNCH("k_cal_gather32", k_cal_gather32, dim3(blocks), dim3(256), 0, (hipStream_t)0, d_a.p, n, table, per_thread, d_sink.p);
    *bytes_out = n * per_thread * 32;
  } else {
    GFS_LAUNCH("k_cal_stream_write", k_cal_stream_write, dim3(blocks), dim3(256), 0, (hipStream_t)0, d_a.p, n);
    *bytes_out = n * 32;
  }
  GFS_HIP(hipDeviceSynchronize());
  return GFS_OK;
}

}  // extern "C"
