// Brute-force 256-bit Hamming matcher for gfx950.
//
// Replaces cv::BFMatcher(cv::NORM_HAMMING).match(d1, d2, matches) at the reference's
// src/ORBmatcher.cc:755-756 / 805-806 / 888-889 and ORBmatcher::DescriptorDistance
// (src/ORBmatcher.cc:2536-2550).  Integer work, bit-exact by construction.
//
// Kernel shape (CDNA4): one workgroup = 4 waves x 64 lanes handles 64 query rows; lane q of every wave
// owns query row q (descriptor held in 4 x u64 VGPR pairs).  The train set streams through LDS in
// tiles of 256 rows (8 KiB, coalesced 16-byte loads); wave w scans rows [64w, 64w+64) of each tile, all
// lanes reading the same LDS address (broadcast, conflict-free).  Each lane keeps (best distance,
// lowest index) with a strict '<' update in increasing train order; the four waves' partial results are
// merged through LDS with the packed key (dist << 20 | idx) so the lowest train index wins ties,
// exactly like OpenCV's batchDistance.
#include "gfs_common.hpp"

namespace {

constexpr int kQPerBlock = 64;
constexpr int kTile = 256;

__global__ __launch_bounds__(256) void k_bf_hamming(const uint8_t* __restrict__ query, const int* __restrict__ nq_arr,
                                                    const uint8_t* __restrict__ train, const int* __restrict__ nt_arr,
                                                    int stride_rows, int* __restrict__ out_idx,
                                                    int* __restrict__ out_dist) {
  __shared__ uint4 tile[kTile * 2];  // 256 rows x 32 B
  __shared__ unsigned int part[4][kQPerBlock];
  const int b = blockIdx.y;
  const int nq = nq_arr[b], nt = nt_arr[b];
  const int q0 = blockIdx.x * kQPerBlock;
  if (q0 >= nq) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = q0 + lane;
  const uint8_t* qb = query + (size_t)b * stride_rows * 32;
  const uint8_t* tb = train + (size_t)b * stride_rows * 32;

  unsigned long long qa[4] = {0, 0, 0, 0};
  if (q < nq) {
    const uint4* qp = reinterpret_cast<const uint4*>(qb + (size_t)q * 32);
    uint4 lo = qp[0], hi = qp[1];
    qa[0] = (unsigned long long)lo.x | ((unsigned long long)lo.y << 32);
    qa[1] = (unsigned long long)lo.z | ((unsigned long long)lo.w << 32);
    qa[2] = (unsigned long long)hi.x | ((unsigned long long)hi.y << 32);
    qa[3] = (unsigned long long)hi.z | ((unsigned long long)hi.w << 32);
  }
  int best = 0x7fffffff, best_j = -1;
  for (int t0 = 0; t0 < nt; t0 += kTile) {
    __syncthreads();
    // 256 rows x 2 uint4: thread i loads uint4 #i and #i+256 (coalesced)
    const uint4* src = reinterpret_cast<const uint4*>(tb + (size_t)t0 * 32);
    const int rows = min(kTile, nt - t0);
    for (int i = threadIdx.x; i < rows * 2; i += 256) tile[i] = src[i];
    __syncthreads();
    const int r0 = wave * 64, r1 = min(r0 + 64, rows);
    for (int r = r0; r < r1; ++r) {
      uint4 lo = tile[2 * r], hi = tile[2 * r + 1];
      unsigned long long t0w = (unsigned long long)lo.x | ((unsigned long long)lo.y << 32);
      unsigned long long t1w = (unsigned long long)lo.z | ((unsigned long long)lo.w << 32);
      unsigned long long t2w = (unsigned long long)hi.x | ((unsigned long long)hi.y << 32);
      unsigned long long t3w = (unsigned long long)hi.z | ((unsigned long long)hi.w << 32);
      int d = __popcll(qa[0] ^ t0w) + __popcll(qa[1] ^ t1w) + __popcll(qa[2] ^ t2w) + __popcll(qa[3] ^ t3w);
      if (d < best) {
        best = d;
        best_j = t0 + r;
      }
    }
  }
  // merge the 4 waves: key = dist (<=256, 9 bits) << 20 | idx (< 2^20)
  part[wave][lane] = best_j < 0 ? 0xffffffffu : ((unsigned int)best << 20) | (unsigned int)best_j;
  __syncthreads();
  if (wave == 0 && q < nq) {
    unsigned int k = min(min(part[0][lane], part[1][lane]), min(part[2][lane], part[3][lane]));
    int* oi = out_idx + (size_t)b * stride_rows;
    int* od = out_dist + (size_t)b * stride_rows;
    if (k == 0xffffffffu) {
      oi[q] = -1;
      od[q] = 0x7fffffff;
    } else {
      oi[q] = (int)(k & 0xfffffu);
      od[q] = (int)(k >> 20);
    }
  }
}

}  // namespace

struct gfs_matcher {
  int device, max_q, max_t, max_b;
  hipStream_t stream;
  gfs::DevBuf<uint8_t> d_q, d_t;
  gfs::DevBuf<int> d_nq, d_nt, d_idx, d_dist;
  std::mutex mu;
};

extern "C" {

// ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:2536-2550): same value, computed with popcount.
int gfs_hamming256(const uint8_t* a, const uint8_t* b) {
  int d = 0;
  for (int i = 0; i < 4; i++) {
    uint64_t x, y;
    memcpy(&x, a + 8 * i, 8);
    memcpy(&y, b + 8 * i, 8);
    d += __builtin_popcountll(x ^ y);
  }
  return d;
}

int gfs_matcher_create(int device, int max_query, int max_train, int max_batch, gfs_matcher** out) {
  GFS_REQUIRE(out && max_query > 0 && max_train > 0 && max_batch > 0, GFS_ERR_INVALID_ARG,
              "gfs_matcher_create: invalid argument");
  GFS_REQUIRE(max_train < (1 << 20), GFS_ERR_UNSUPPORTED, "gfs_matcher_create: max_train must be < 2^20");
  if (!gfs::device_ok(device)) return GFS_ERR_NO_DEVICE;
  GFS_HIP(hipSetDevice(device));
  gfs_matcher* h = new gfs_matcher;
  h->device = device;
  h->max_q = max_query;
  h->max_t = max_train;
  h->max_b = max_batch;
  int rc;
  GFS_HIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  const size_t rows = (size_t)std::max(max_query, max_train);
  if ((rc = h->d_q.alloc(rows * 32)) || (rc = h->d_t.alloc(rows * 32)) || (rc = h->d_nq.alloc(1)) ||
      (rc = h->d_nt.alloc(1)) || (rc = h->d_idx.alloc(rows)) || (rc = h->d_dist.alloc(rows))) {
    delete h;
    return rc;
  }
  *out = h;
  return GFS_OK;
}

void gfs_matcher_destroy(gfs_matcher* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  (void)hipStreamDestroy(h->stream);
  delete h;
}

int gfs_bf_match_hamming_batch_device(gfs_matcher* h, const void* dev_query, const void* dev_nq, const void* dev_train,
                                      const void* dev_nt, int B, int stride_rows, void* dev_train_idx, void* dev_dist,
                                      void* stream) {
  GFS_REQUIRE(h && dev_query && dev_nq && dev_train && dev_nt && dev_train_idx && dev_dist && B > 0 && stride_rows > 0,
              GFS_ERR_INVALID_ARG, "gfs_bf_match_hamming_batch_device: invalid argument");
  GFS_REQUIRE(stride_rows < (1 << 20), GFS_ERR_UNSUPPORTED, "stride_rows must be < 2^20");
  GFS_HIP(hipSetDevice(h->device));
  hipStream_t s = stream ? (hipStream_t)stream : h->stream;
  dim3 grid(gfs::div_up(stride_rows, kQPerBlock), B);
  GFS_LAUNCH("k_bf_hamming", k_bf_hamming, grid, dim3(256), 0, s, (const uint8_t*)dev_query, (const int*)dev_nq,
             (const uint8_t*)dev_train, (const int*)dev_nt, stride_rows, (int*)dev_train_idx, (int*)dev_dist);
  return GFS_OK;
}

int gfs_bf_match_hamming(gfs_matcher* h, const uint8_t* query, int nq, const uint8_t* train, int nt,
                         int32_t* train_idx, int32_t* dist) {
  GFS_REQUIRE(h && nq >= 0 && nt >= 0, GFS_ERR_INVALID_ARG, "gfs_bf_match_hamming: invalid argument");
  if (nq == 0 || nt == 0) return 0;  // OpenCV: empty train set -> no matches
  GFS_REQUIRE(query && train && train_idx && dist, GFS_ERR_INVALID_ARG, "gfs_bf_match_hamming: NULL buffer");
  GFS_REQUIRE(nq <= h->max_q && nt <= h->max_t, GFS_ERR_CAPACITY, "gfs_bf_match_hamming: %d x %d exceeds handle capacity %d x %d",
              nq, nt, h->max_q, h->max_t);
  std::lock_guard<std::mutex> lk(h->mu);
  GFS_HIP(hipSetDevice(h->device));
  hipStream_t s = h->stream;
  const int stride_rows = std::max(h->max_q, h->max_t);
  GFS_HIP(hipMemcpyAsync(h->d_q.p, query, (size_t)nq * 32, hipMemcpyHostToDevice, s));
  GFS_HIP(hipMemcpyAsync(h->d_t.p, train, (size_t)nt * 32, hipMemcpyHostToDevice, s));
  GFS_HIP(hipMemcpyAsync(h->d_nq.p, &nq, sizeof(int), hipMemcpyHostToDevice, s));
  GFS_HIP(hipMemcpyAsync(h->d_nt.p, &nt, sizeof(int), hipMemcpyHostToDevice, s));
  int rc = gfs_bf_match_hamming_batch_device(h, h->d_q.p, h->d_nq.p, h->d_t.p, h->d_nt.p, 1, stride_rows, h->d_idx.p,
                                             h->d_dist.p, s);
  if (rc) return rc;
  GFS_HIP(hipMemcpyAsync(train_idx, h->d_idx.p, (size_t)nq * sizeof(int), hipMemcpyDeviceToHost, s));
  GFS_HIP(hipMemcpyAsync(dist, h->d_dist.p, (size_t)nq * sizeof(int), hipMemcpyDeviceToHost, s));
  GFS_HIP(hipStreamSynchronize(s));
  return nq;
}

}  // extern "C"
