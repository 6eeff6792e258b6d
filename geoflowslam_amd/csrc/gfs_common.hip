// Error reporting, device probing, timers and the per-kernel profiler of libgfs_hip.so.
#include "gfs_common.hpp"

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

}  // extern "C"
