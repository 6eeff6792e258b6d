// Frame-level helpers adjacent to the hot path: they keep the GICP input and the RGB-D
// "stereo" coordinates on the device between ORB extraction and registration.
//   Frame::ConvertDepthToPointCloud   reference src/Frame.cc:590-623  -> k_depth_to_cloud (ordered compaction)
//   Frame::ComputeStereoFromRGBD      reference src/Frame.cc:1314-1332 -> k_stereo_from_rgbd
// Float arithmetic is issued op by op (__fsub_rn / __fmul_rn / __fdiv_rn): bit-exact with the reference expressions.
#include <memory>

#include "gfs_common.hpp"

namespace {

// One 1024-thread workgroup per frame walks the stride grid in raster order, 1024 samples per iteration;
// a block-wide exclusive scan of the validity flags gives every surviving point its reference push_back position.
__global__ __launch_bounds__(1024) void k_depth_to_cloud(const float* __restrict__ depth, size_t frame_stride, int rows,
                                                         int cols, int pitch, int ds, float fx, float fy, float cx, float cy,
                                                         float4* __restrict__ out, int stride_pts, int* __restrict__ counts) {
  __shared__ int s_wave[16];
  __shared__ int s_carry;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* D = depth + (size_t)b * frame_stride;
  float4* O = out + (size_t)b * stride_pts;
  const int gw = (cols + ds - 1) / ds, gh = (rows + ds - 1) / ds, total = gw * gh;
  if (tid == 0) s_carry = 0;
  __syncthreads();
  for (int t0 = 0; t0 < total; t0 += 1024) {
    const int i = t0 + tid;
    bool ok = false;
    float d = 0.f;
    int u = 0, v = 0;
    if (i < total) {
      v = (i / gw) * ds;
      u = (i - (i / gw) * gw) * ds;
      d = D[(size_t)v * pitch + u];
      ok = d > 0.0f && d < 10.0f;  // if (depth > 0.0 && depth < 10.0)
    }
    const unsigned long long bal = __ballot(ok);
    const int wrank = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) s_wave[wave] = __popcll(bal);
    __syncthreads();
    int base = s_carry, tot = 0;
    for (int w = 0; w < 16; w++) {
      if (w < wave) base += s_wave[w];
      tot += s_wave[w];
    }
    if (ok) {
      const int pos = base + wrank;
      if (pos < stride_pts) {
        const float x = __fdiv_rn(__fmul_rn(__fsub_rn((float)u, cx), d), fx);  // (u - cx) * depth / fx
        const float y = __fdiv_rn(__fmul_rn(__fsub_rn((float)v, cy), d), fy);
        O[pos] = make_float4(x, y, d, 1.0f);
      }
    }
    __syncthreads();
    if (tid == 0) s_carry += tot;
    __syncthreads();
  }
  if (tid == 0) counts[b] = s_carry;  // may exceed stride_pts: the host entry reports GFS_ERR_CAPACITY
}

// imDepth.convertTo(imDepth, CV_32F, mDepthMapFactor) on a CV_16U sensor image (reference src/Tracking.cc:1622-1623; OpenCV
// cvtScale16u32f: float(src) * float(alpha) + 0 in single precision, one rounding with or without FMA)
__global__ __launch_bounds__(256) void k_depth_u16_to_f32(const unsigned short* __restrict__ src, size_t n, float factor,
                                                          float* __restrict__ dst) {
  const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i + 3 < n) {  // (the batches are rows x cols x B: eight-byte aligned groups of four)
    const ushort4 v = *reinterpret_cast<const ushort4*>(src + i);
    *reinterpret_cast<float4*>(dst + i) = make_float4(__fmul_rn((float)v.x, factor), __fmul_rn((float)v.y, factor),
                                                       __fmul_rn((float)v.z, factor), __fmul_rn((float)v.w, factor));
  } else {
    for (size_t k = i; k < n; k++) dst[k] = __fmul_rn((float)src[k], factor);
  }
}

__global__ void k_stereo_from_rgbd(const gfs_keypoint* __restrict__ kps, const float* __restrict__ kps_un_x,
                                   const int* __restrict__ n_arr, int kp_stride, const float* __restrict__ depth,
                                   size_t frame_stride, int pitch, float bf, float* __restrict__ u_right,
                                   float* __restrict__ v_depth) {
  const int b = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_arr[b]) return;
  const gfs_keypoint kp = kps[(size_t)b * kp_stride + i];
  const float xun = kps_un_x ? kps_un_x[(size_t)b * kp_stride + i] : kp.x;
  // imDepth.at<float>(v, u) with float v, u: implicit float -> int conversion (truncation)
  const float d = depth[(size_t)b * frame_stride + (size_t)(int)kp.y * pitch + (int)kp.x];
  float ur = -1.f, vd = -1.f;
  if (d > 0) {
    vd = d;
    ur = __fsub_rn(xun, __fdiv_rn(bf, d));  // kpU.pt.x - mbf / d
  }
  u_right[(size_t)b * kp_stride + i] = ur;
  v_depth[(size_t)b * kp_stride + i] = vd;
}

}  // namespace

struct gfs_frame {
  int device, max_rows, max_cols, max_kp;
  hipStream_t stream;
  std::mutex mu;
  gfs::DevBuf<float> d_depth, d_unx, d_ur, d_vd;
  gfs::DevBuf<float4> d_cloud;
  gfs::DevBuf<gfs_keypoint> d_kps;
  gfs::DevBuf<int> d_n;
  int res_rows = 0, res_cols = 0;  // size of the depth map resident in d_depth (gfs_frame_rgbd)
};

// Error paths of the host-pointer entries must not return while asynchronous copies into the CALLER's buffers (or out of a
// stack variable) are still in flight on the handle's stream: armed after the first asynchronous operation, disarmed by the
// regular synchronisation.
struct StreamDrain {
  hipStream_t s;
  bool armed = true;
  ~StreamDrain() {
    if (armed) (void)hipStreamSynchronize(s);
  }
};

extern "C" {

int gfs_frame_create(int device, int max_rows, int max_cols, int max_keypoints, gfs_frame** out) {
  GFS_REQUIRE(out && max_rows > 0 && max_cols > 0 && max_keypoints > 0, GFS_ERR_INVALID_ARG, "gfs_frame_create: invalid argument");
  if (!gfs::device_ok(device)) return GFS_ERR_NO_DEVICE;
  GFS_HIP(hipSetDevice(device));
  std::unique_ptr<gfs_frame> h(new gfs_frame);
  h->device = device;
  h->max_rows = max_rows;
  h->max_cols = max_cols;
  h->max_kp = max_keypoints;
  GFS_HIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  int rc = 0;
#define A(x) if (!rc) rc = (x)
  A(h->d_depth.alloc((size_t)max_rows * max_cols));
  A(h->d_cloud.alloc((size_t)max_rows * max_cols));
  A(h->d_kps.alloc(max_keypoints));
  A(h->d_unx.alloc(max_keypoints));
  A(h->d_ur.alloc(max_keypoints));
  A(h->d_vd.alloc(max_keypoints));
  A(h->d_n.alloc(2));  // [0] cloud points, [1] key-points of a gfs_frame_rgbd call
#undef A
  if (rc) return rc;
  *out = h.release();
  return GFS_OK;
}

void gfs_frame_destroy(gfs_frame* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  (void)hipStreamSynchronize(h->stream);
  (void)hipStreamDestroy(h->stream);
  delete h;
}

int gfs_depth_to_cloud_batch_device(gfs_frame* h, const void* dev_depth, int B, int rows, int cols, int downsample, float fx,
                                    float fy, float cx, float cy, void* dev_out_xyzw, int stride_pts, void* dev_counts,
                                    void* stream) {
  GFS_REQUIRE(h && dev_depth && dev_out_xyzw && dev_counts && B > 0 && rows > 0 && cols > 0 && downsample > 0 && stride_pts > 0,
              GFS_ERR_INVALID_ARG, "gfs_depth_to_cloud_batch_device: invalid argument");
  GFS_HIP(hipSetDevice(h->device));
  hipStream_t s = stream ? (hipStream_t)stream : h->stream;
  GFS_LAUNCH("k_depth_to_cloud", k_depth_to_cloud, dim3(B), dim3(1024), 0, s, (const float*)dev_depth, (size_t)rows * cols, rows,
             cols, cols, downsample, fx, fy, cx, cy, (float4*)dev_out_xyzw, stride_pts, (int*)dev_counts);
  return GFS_OK;
}

int gfs_depth_convert_u16_batch_device(gfs_frame* h, const void* dev_depth_u16, int B, int rows, int cols, float factor,
                                       void* dev_depth_f32, void* stream) {
  GFS_REQUIRE(h && dev_depth_u16 && dev_depth_f32 && B > 0 && rows > 0 && cols > 0, GFS_ERR_INVALID_ARG,
              "gfs_depth_convert_u16_batch_device: invalid argument");
  GFS_REQUIRE(((uintptr_t)dev_depth_u16 & 7) == 0 && ((uintptr_t)dev_depth_f32 & 15) == 0, GFS_ERR_INVALID_ARG,
              "gfs_depth_convert_u16_batch_device: buffers must be 8 / 16 byte aligned");
  GFS_HIP(hipSetDevice(h->device));
  hipStream_t s = stream ? (hipStream_t)stream : h->stream;
  const size_t n = (size_t)B * rows * cols;
  GFS_LAUNCH("k_depth_u16_to_f32", k_depth_u16_to_f32, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, s,
             (const unsigned short*)dev_depth_u16, n, factor, (float*)dev_depth_f32);
  return GFS_OK;
}

int gfs_depth_to_cloud(gfs_frame* h, const float* depth, int rows, int cols, int stride_elems, int downsample, float fx,
                       float fy, float cx, float cy, float* out_xyzw, int cap, int* n) {
  GFS_REQUIRE(h && n, GFS_ERR_INVALID_ARG, "gfs_depth_to_cloud: NULL argument");
  *n = 0;
  if (!depth || rows <= 0 || cols <= 0) return GFS_OK;  // "Depth image is empty": the reference returns without points
  GFS_REQUIRE(downsample > 0 && stride_elems >= cols && rows <= h->max_rows && cols <= h->max_cols, GFS_ERR_INVALID_ARG,
              "gfs_depth_to_cloud: invalid geometry");
  std::lock_guard<std::mutex> lk(h->mu);
  h->res_rows = h->res_cols = 0;  // d_depth is about to be overwritten (under the lock: gfs_frame_rgbd(depth = NULL) reads these)
  GFS_HIP(hipSetDevice(h->device));
  hipStream_t s = h->stream;
  StreamDrain drain{s};
  GFS_HIP(hipMemcpy2DAsync(h->d_depth.p, (size_t)cols * 4, depth, (size_t)stride_elems * 4, (size_t)cols * 4, rows,
                           hipMemcpyHostToDevice, s));
  const int maxpts = (int)h->d_cloud.n;
  int rc = gfs_depth_to_cloud_batch_device(h, h->d_depth.p, 1, rows, cols, downsample, fx, fy, cx, cy, h->d_cloud.p, maxpts,
                                           h->d_n.p, s);
  if (rc) return rc;
  int cnt = 0;
  GFS_HIP(hipMemcpyAsync(&cnt, h->d_n.p, sizeof(int), hipMemcpyDeviceToHost, s));
  GFS_HIP(hipStreamSynchronize(s));
  drain.armed = false;
  *n = cnt;
  GFS_REQUIRE(cnt <= cap, GFS_ERR_CAPACITY, "gfs_depth_to_cloud: %d points exceed caller capacity %d", cnt, cap);
  if (cnt && out_xyzw) GFS_HIP(hipMemcpy(out_xyzw, h->d_cloud.p, (size_t)cnt * 16, hipMemcpyDeviceToHost));
  return GFS_OK;
}

int gfs_stereo_from_rgbd_batch_device(gfs_frame* h, const void* dev_kps, const void* dev_kps_un_x, const void* dev_counts,
                                      int B, int kp_stride, const void* dev_depth, int rows, int cols, float bf,
                                      void* dev_u_right, void* dev_depth_out, void* stream) {
  GFS_REQUIRE(h && dev_kps && dev_counts && dev_depth && dev_u_right && dev_depth_out && B > 0 && kp_stride > 0 && rows > 0 &&
                  cols > 0,
              GFS_ERR_INVALID_ARG, "gfs_stereo_from_rgbd_batch_device: invalid argument");
  GFS_HIP(hipSetDevice(h->device));
  hipStream_t s = stream ? (hipStream_t)stream : h->stream;
  GFS_LAUNCH("k_stereo_from_rgbd", k_stereo_from_rgbd, dim3(gfs::div_up(kp_stride, 256), B), dim3(256), 0, s,
             (const gfs_keypoint*)dev_kps, (const float*)dev_kps_un_x, (const int*)dev_counts, kp_stride, (const float*)dev_depth,
             (size_t)rows * cols, cols, bf, (float*)dev_u_right, (float*)dev_depth_out);
  return GFS_OK;
}

int gfs_stereo_from_rgbd(gfs_frame* h, const gfs_keypoint* kps, const float* kps_un_x, int n, const float* depth, int rows,
                         int cols, int stride_elems, float bf, float* u_right, float* depth_out) {
  GFS_REQUIRE(h && n >= 0, GFS_ERR_INVALID_ARG, "gfs_stereo_from_rgbd: invalid argument");
  if (n == 0) return GFS_OK;
  GFS_REQUIRE(kps && depth && u_right && depth_out && rows > 0 && cols > 0 && stride_elems >= cols && rows <= h->max_rows &&
                  cols <= h->max_cols && n <= h->max_kp,
              GFS_ERR_INVALID_ARG, "gfs_stereo_from_rgbd: invalid argument or capacity");
  std::lock_guard<std::mutex> lk(h->mu);
  h->res_rows = h->res_cols = 0;  // (under the lock, as in gfs_depth_to_cloud)
  GFS_HIP(hipSetDevice(h->device));
  hipStream_t s = h->stream;
  StreamDrain drain{s};
  GFS_HIP(hipMemcpy2DAsync(h->d_depth.p, (size_t)cols * 4, depth, (size_t)stride_elems * 4, (size_t)cols * 4, rows,
                           hipMemcpyHostToDevice, s));
  GFS_HIP(hipMemcpyAsync(h->d_kps.p, kps, (size_t)n * sizeof(gfs_keypoint), hipMemcpyHostToDevice, s));
  if (kps_un_x) GFS_HIP(hipMemcpyAsync(h->d_unx.p, kps_un_x, (size_t)n * 4, hipMemcpyHostToDevice, s));
  GFS_HIP(hipMemcpyAsync(h->d_n.p, &n, sizeof(int), hipMemcpyHostToDevice, s));
  int rc = gfs_stereo_from_rgbd_batch_device(h, h->d_kps.p, kps_un_x ? h->d_unx.p : nullptr, h->d_n.p, 1, h->max_kp, h->d_depth.p,
                                             rows, cols, bf, h->d_ur.p, h->d_vd.p, s);
  if (rc) return rc;
  GFS_HIP(hipMemcpyAsync(u_right, h->d_ur.p, (size_t)n * 4, hipMemcpyDeviceToHost, s));
  GFS_HIP(hipMemcpyAsync(depth_out, h->d_vd.p, (size_t)n * 4, hipMemcpyDeviceToHost, s));
  GFS_HIP(hipStreamSynchronize(s));
  drain.armed = false;
  return GFS_OK;
}

// The RGB-D tail of the Frame constructor in one call (src/Frame.cc: ComputeStereoFromRGBD(imDepth) :1314-1332, then
// ConvertDepthToPointCloud :590-623): the depth map crosses PCIe ONCE, both kernels run behind it, one synchronisation.  The cloud
// stays on the device for the registration (dev_cloud / dev_count are what gfs_gicp_align[_next]_batch_device take, stride
// *cloud_stride points); out_xyzw may be NULL when the caller does not need a host copy.
int gfs_frame_rgbd(gfs_frame* h, const gfs_keypoint* kps, const float* kps_un_x, int n, const float* depth, int rows, int cols,
                   int stride_elems, float bf, int downsample, float fx, float fy, float cx, float cy, float* u_right,
                   float* depth_out, float* out_xyzw, int cap, int* n_cloud, void** dev_cloud, void** dev_count, int* cloud_stride) {
  GFS_REQUIRE(h && n >= 0 && n_cloud, GFS_ERR_INVALID_ARG, "gfs_frame_rgbd: invalid argument");
  *n_cloud = 0;
  // depth == NULL: the depth map of the previous call on this handle (same rows x cols) is still on the device -- a caller that
  // overlaps the ORB extraction with the registration asks for the cloud first and for the stereo coordinates once the key-points
  // exist; downsample <= 0: no cloud in this call.
  GFS_REQUIRE(rows > 0 && cols > 0 && (depth == nullptr || stride_elems >= cols) && rows <= h->max_rows && cols <= h->max_cols &&
                  n <= h->max_kp && (n == 0 || (kps && u_right && depth_out)),
              GFS_ERR_INVALID_ARG, "gfs_frame_rgbd: invalid argument or capacity");
  std::lock_guard<std::mutex> lk(h->mu);
  GFS_REQUIRE(depth != nullptr || (h->res_rows == rows && h->res_cols == cols), GFS_ERR_INVALID_ARG,
              "gfs_frame_rgbd: no resident %dx%d depth map on this handle", cols, rows);
  GFS_HIP(hipSetDevice(h->device));
  hipStream_t s = h->stream;
  StreamDrain drain{s};
  if (depth) {
    h->res_rows = h->res_cols = 0;
    GFS_HIP(hipMemcpy2DAsync(h->d_depth.p, (size_t)cols * 4, depth, (size_t)stride_elems * 4, (size_t)cols * 4, rows,
                             hipMemcpyHostToDevice, s));
    h->res_rows = rows;
    h->res_cols = cols;
  }
  if (n) {
    GFS_HIP(hipMemcpyAsync(h->d_kps.p, kps, (size_t)n * sizeof(gfs_keypoint), hipMemcpyHostToDevice, s));
    if (kps_un_x) GFS_HIP(hipMemcpyAsync(h->d_unx.p, kps_un_x, (size_t)n * 4, hipMemcpyHostToDevice, s));
    GFS_HIP(hipMemcpyAsync(h->d_n.p + 1, &n, sizeof(int), hipMemcpyHostToDevice, s));
    const int rc = gfs_stereo_from_rgbd_batch_device(h, h->d_kps.p, kps_un_x ? h->d_unx.p : nullptr, h->d_n.p + 1, 1, h->max_kp,
                                                     h->d_depth.p, rows, cols, bf, h->d_ur.p, h->d_vd.p, s);
    if (rc) return rc;
    GFS_HIP(hipMemcpyAsync(u_right, h->d_ur.p, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    GFS_HIP(hipMemcpyAsync(depth_out, h->d_vd.p, (size_t)n * 4, hipMemcpyDeviceToHost, s));
  }
  const int maxpts = (int)h->d_cloud.n;
  int cnt = 0;
  if (downsample > 0) {
    const int rc = gfs_depth_to_cloud_batch_device(h, h->d_depth.p, 1, rows, cols, downsample, fx, fy, cx, cy, h->d_cloud.p, maxpts,
                                                   h->d_n.p, s);
    if (rc) return rc;
    GFS_HIP(hipMemcpyAsync(&cnt, h->d_n.p, sizeof(int), hipMemcpyDeviceToHost, s));
  }
  GFS_HIP(hipStreamSynchronize(s));
  drain.armed = false;
  *n_cloud = cnt;
  if (dev_cloud) *dev_cloud = h->d_cloud.p;
  if (dev_count) *dev_count = h->d_n.p;
  if (cloud_stride) *cloud_stride = maxpts;
  if (out_xyzw) {
    GFS_REQUIRE(cnt <= cap, GFS_ERR_CAPACITY, "gfs_frame_rgbd: %d points exceed caller capacity %d", cnt, cap);
    if (cnt) GFS_HIP(hipMemcpy(out_xyzw, h->d_cloud.p, (size_t)cnt * 16, hipMemcpyDeviceToHost));
  }
  return GFS_OK;
}

}  // extern "C"
