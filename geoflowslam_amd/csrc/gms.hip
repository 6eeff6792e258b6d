// gms_matcher(...).GetInlierMask(mask, WithScale = false, WithRotation = false) (reference, vendored:
// Thirdparty/GMS/include/gms_matcher.h:43-60, 289-301, 356-455) on MI355X — the grid-based motion-statistics filter the
// reference runs right after every cv::BFMatcher::match of the hot path (src/ORBmatcher.cc:761-762, 812-813, 893-894).
//
// One 256-thread workgroup per (frame pair, shifted grid): the four grids vote independently, k_gms_merge ORs their verdicts and
// counts the inliers.  The reference builds a dense 400 x 400 vote matrix per shifted grid; with
// ~1000 matches it is almost empty, so the votes are kept as a CSR list of right-cell indices per left cell in LDS:
// a left cell's best partner (first maximum over ascending right index) and its 3 x 3 neighbourhood score are counted
// from those lists.  Integer work, bit-exact; the float / double grid arithmetic follows the header's expressions.
#include <memory>
#include <mutex>

#include "gfs_common.hpp"

namespace {

constexpr int kGmsW = 20, kGmsH = 20, kGmsCells = kGmsW * kGmsH;
constexpr int kGmsThreads = 256;
constexpr int kGmsMaxMatches = 8192;

struct GmsPair {
  int n1, n2, n_matches, w1, h1, w2, h2, implicit_query;  // implicit_query: match i = (i, train_idx[i])
};

__device__ __forceinline__ int gms_left_index(float px, float py, int type) {  // GetGridIndexLeft :146-176
  int x, y;
  if (type == 1) {
    x = (int)floorf(px * kGmsW);
    y = (int)floorf(py * kGmsH);
  } else if (type == 2) {
    x = (int)floor((double)(px * kGmsW) + 0.5);
    y = (int)floorf(py * kGmsH);
  } else if (type == 3) {
    x = (int)floorf(px * kGmsW);
    y = (int)floor((double)(py * kGmsH) + 0.5);
  } else {
    x = (int)floor((double)(px * kGmsW) + 0.5);
    y = (int)floor((double)(py * kGmsH) + 0.5);
  }
  if (x >= kGmsW || y >= kGmsH) return -1;
  return x + y * kGmsW;
}

__device__ __forceinline__ int gms_nb(int idx, int j) {  // GetNB9 :190-211, entry j = xi + 1 + 3 (yi + 1)
  const int xi = j % 3 - 1, yi = j / 3 - 1;
  const int xx = idx % kGmsW + xi, yy = idx / kGmsW + yi;
  if (xx < 0 || xx >= kGmsW || yy < 0 || yy >= kGmsH) return -1;
  return xx + yy * kGmsW;
}

__global__ __launch_bounds__(kGmsThreads) void k_gms(const GmsPair* __restrict__ pairs, GmsPair common, const gfs_keypoint* __restrict__ kp1_all,
                                                     const gfs_keypoint* __restrict__ kp2_all, const int* __restrict__ n1_dev,
                                                     const int* __restrict__ n2_dev, int kp_stride, const int* __restrict__ q_all,
                                                     const int* __restrict__ t_all, int m_stride, uint8_t* __restrict__ vote_all) {
  __shared__ short s_l[kGmsMaxMatches], s_r[kGmsMaxMatches];
  __shared__ unsigned short s_items[kGmsMaxMatches];  // right-cell index of the valid votes, grouped by left cell
  __shared__ int s_cnt[kGmsCells], s_start[kGmsCells + 1], s_cur[kGmsCells];
  __shared__ int s_pair[kGmsCells];
  __shared__ int s_scan[kGmsThreads / 64];
  const int f = blockIdx.y, type = blockIdx.x + 1, tid = threadIdx.x;  // one workgroup per (frame pair, shifted grid)
  GmsPair P = pairs ? pairs[f] : common;
  if (n1_dev) {  // device-resident batch: counts come from the extractor's device results
    P.n1 = n1_dev[f];
    P.n2 = n2_dev[f];
    P.n_matches = P.n2 > 0 ? min(P.n1, kGmsMaxMatches) : 0;  // BFMatcher returns one match per query unless the train set is empty
  }
  const gfs_keypoint* kp1 = kp1_all + (size_t)f * kp_stride;
  const gfs_keypoint* kp2 = kp2_all + (size_t)f * kp_stride;
  const int* qi = q_all ? q_all + (size_t)f * m_stride : nullptr;
  const int* ti = t_all + (size_t)f * m_stride;
  uint8_t* vote = vote_all + ((size_t)f * 4 + (type - 1)) * m_stride;  // this grid's verdict on every match; k_gms_merge ORs the four
  const int M = P.n_matches;
  {
    for (int c = tid; c < kGmsCells; c += kGmsThreads) {
      s_cnt[c] = 0;
      s_pair[c] = -1;
    }
    __syncthreads();
    // AssignMatchPairs :356-383
    for (int i = tid; i < M; i += kGmsThreads) {
      const int q = P.implicit_query ? i : qi[i];
      const gfs_keypoint a = kp1[q];
      const float lx = a.x / P.w1, ly = a.y / P.h1;  // NormalizePoints: float / int
      const int l = gms_left_index(lx, ly, type);
      s_l[i] = (short)l;
      int r;
      {  // the right grid is not shifted: the same index for all four types (the reference computes it once, for type 1)
        const gfs_keypoint b = kp2[ti[i]];
        const float rx = b.x / P.w2, ry = b.y / P.h2;
        r = (int)floorf(rx * kGmsW) + (int)floorf(ry * kGmsH) * kGmsW;  // GetGridIndexRight :178-183 (no range check)
        r = max(min(r, 32767), -32768);
        s_r[i] = (short)r;
      }
      if (l >= 0 && r >= 0 && r < kGmsCells) atomicAdd(&s_cnt[l], 1);
    }
    __syncthreads();
    {  // exclusive scan of the 400 cell counts
      const int c0 = 2 * tid, c1 = 2 * tid + 1, lane = tid & 63, wave = tid >> 6;
      const int a = c0 < kGmsCells ? s_cnt[c0] : 0, b = c1 < kGmsCells ? s_cnt[c1] : 0;
      int incl = a + b;  // wave scan + four wave totals: two barriers instead of sixteen
      for (int ofs = 1; ofs < 64; ofs <<= 1) {
        const int v = __shfl_up(incl, ofs, 64);
        if (lane >= ofs) incl += v;
      }
      if (lane == 63) s_scan[wave] = incl;
      __syncthreads();
      int wbase = 0, total = 0;
      for (int w = 0; w < kGmsThreads / 64; w++) {
        if (w < wave) wbase += s_scan[w];
        total += s_scan[w];
      }
      const int base = wbase + incl - (a + b);
      if (c0 < kGmsCells) {
        s_start[c0] = base;
        s_cur[c0] = base;
      }
      if (c1 < kGmsCells) {
        s_start[c1] = base + a;
        s_cur[c1] = base + a;
      }
      if (tid == kGmsThreads - 1) s_start[kGmsCells] = total;
    }
    __syncthreads();
    for (int i = tid; i < M; i += kGmsThreads) {
      const int l = s_l[i], r = s_r[i];
      if (l >= 0 && r >= 0 && r < kGmsCells) s_items[atomicAdd(&s_cur[l], 1)] = (unsigned short)r;
    }
    __syncthreads();
    // VerifyCellPairs(1) :386-421, one thread per left cell
    for (int c = tid; c < kGmsCells; c += kGmsThreads) {
      const int b0 = s_start[c], b1 = s_start[c + 1];
      if (b1 == b0) continue;  // empty row: stays -1
      int best = -1, best_n = 0;
      for (int a = b0; a < b1; a++) {  // first maximum over ascending right index
        const int r = s_items[a];
        int n = 0;
        for (int e = b0; e < b1; e++) n += s_items[e] == r ? 1 : 0;
        if (n > best_n || (n == best_n && r < best)) {
          best_n = n;
          best = r;
        }
      }
      int score = 0, numpair = 0;
      double thresh = 0;
      for (int j = 0; j < 9; j++) {
        const int ll = gms_nb(c, j), rr = gms_nb(best, j);  // rotation pattern 1 is the identity
        if (ll == -1 || rr == -1) continue;
        for (int e = s_start[ll]; e < s_start[ll + 1]; e++) score += s_items[e] == rr ? 1 : 0;
        thresh += (double)(s_start[ll + 1] - s_start[ll]);  // mNumberPointsInPerCellLeft[ll]
        numpair++;
      }
      thresh = 6 * sqrt(thresh / numpair);  // THRESH_FACTOR
      s_pair[c] = (double)score < thresh ? -2 : best;
    }
    __syncthreads();
    for (int i = tid; i < M; i += kGmsThreads) {
      const int l = s_l[i];
      // (the reference reads mCellPairs[-1], out of bounds, for l == -1: never equal to a right index)
      vote[i] = l >= 0 && s_pair[l] == (int)s_r[i] ? 1 : 0;
    }
  }
}

// mvbInlierMask |= over the four shifted grids (gms_matcher::run :440-452) and the inlier count
__global__ __launch_bounds__(kGmsThreads) void k_gms_merge(const GmsPair* __restrict__ pairs, GmsPair common, const int* __restrict__ n1_dev,
                                                           const int* __restrict__ n2_dev, int m_stride,
                                                           const uint8_t* __restrict__ vote_all, uint8_t* __restrict__ mask_all,
                                                           int* __restrict__ counts) {
  __shared__ int s_total;
  const int f = blockIdx.x, tid = threadIdx.x;
  GmsPair P = pairs ? pairs[f] : common;
  if (n1_dev) P.n_matches = n2_dev[f] > 0 ? min(n1_dev[f], kGmsMaxMatches) : 0;
  const uint8_t* v = vote_all + (size_t)f * 4 * m_stride;
  uint8_t* mask = mask_all + (size_t)f * m_stride;
  if (tid == 0) s_total = 0;
  __syncthreads();
  int local = 0;
  for (int i = tid; i < P.n_matches; i += kGmsThreads) {
    const uint8_t m = v[i] | v[m_stride + i] | v[2 * m_stride + i] | v[3 * m_stride + i];
    mask[i] = m;
    local += m;
  }
  if (local) atomicAdd(&s_total, local);
  __syncthreads();
  if (tid == 0) counts[f] = s_total;
}

}  // namespace

struct gfs_gms {
  int device, max_kps, max_batch;
  hipStream_t stream;
  std::mutex mu;
  gfs::DevBuf<GmsPair> d_pairs;
  gfs::DevBuf<gfs_keypoint> d_kp1, d_kp2;
  gfs::DevBuf<int> d_q, d_t, d_counts;
  gfs::DevBuf<uint8_t> d_mask, d_vote;
  gfs::PinBuf<GmsPair> h_pairs;
  gfs::PinBuf<gfs_keypoint> h_kp1, h_kp2;
  gfs::PinBuf<int> h_q, h_t, h_counts;
  gfs::PinBuf<uint8_t> h_mask;
};

extern "C" {

int gfs_gms_create(int device, int max_keypoints, int max_batch, gfs_gms** out) {
  GFS_REQUIRE(out && max_keypoints > 0 && max_batch > 0, GFS_ERR_INVALID_ARG, "gfs_gms_create: invalid argument");
  GFS_REQUIRE(max_keypoints <= kGmsMaxMatches, GFS_ERR_UNSUPPORTED, "gfs_gms_create: at most %d key-points / matches per frame", kGmsMaxMatches);
  if (!gfs::device_ok(device)) return GFS_ERR_NO_DEVICE;
  GFS_HIP(hipSetDevice(device));
  std::unique_ptr<gfs_gms> h(new gfs_gms);
  h->device = device;
  h->max_kps = max_keypoints;
  h->max_batch = max_batch;
  GFS_HIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  const size_t K = (size_t)max_keypoints * max_batch, B = max_batch;
  int rc = 0;
#define A(x) if (!rc) rc = (x)
  A(h->d_pairs.alloc(B));
  A(h->d_kp1.alloc(K));
  A(h->d_kp2.alloc(K));
  A(h->d_q.alloc(K));
  A(h->d_t.alloc(K));
  A(h->d_mask.alloc(K));
  A(h->d_vote.alloc(4 * K));
  A(h->d_counts.alloc(B));
  A(h->h_pairs.alloc(B));
  A(h->h_kp1.alloc(K));
  A(h->h_kp2.alloc(K));
  A(h->h_q.alloc(K));
  A(h->h_t.alloc(K));
  A(h->h_mask.alloc(K));
  A(h->h_counts.alloc(B));
#undef A
  if (rc) {
    (void)hipStreamDestroy(h->stream);
    return rc;
  }
  *out = h.release();
  return GFS_OK;
}

void gfs_gms_destroy(gfs_gms* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  (void)hipStreamSynchronize(h->stream);
  (void)hipStreamDestroy(h->stream);
  delete h;
}

int gfs_gms_inlier_mask(gfs_gms* h, const gfs_gms_problem* problems, int B, uint8_t* const* inlier, int32_t* n_inliers) {
  GFS_REQUIRE(h && problems && inlier && n_inliers && B > 0, GFS_ERR_INVALID_ARG, "gfs_gms_inlier_mask: invalid argument");
  GFS_REQUIRE(B <= h->max_batch, GFS_ERR_CAPACITY, "gfs_gms_inlier_mask: batch %d exceeds capacity %d", B, h->max_batch);
  std::lock_guard<std::mutex> lk(h->mu);
  GFS_HIP(hipSetDevice(h->device));
  const int S = h->max_kps;
  for (int f = 0; f < B; f++) {
    const gfs_gms_problem& p = problems[f];
    GFS_REQUIRE(p.n1 >= 0 && p.n1 <= S && p.n2 >= 0 && p.n2 <= S && p.n_matches >= 0 && p.n_matches <= S, GFS_ERR_CAPACITY,
                "gfs_gms_inlier_mask: pair %d (%d / %d key-points, %d matches) exceeds capacity %d", f, p.n1, p.n2, p.n_matches, S);
    GFS_REQUIRE(p.width1 > 0 && p.height1 > 0 && p.width2 > 0 && p.height2 > 0, GFS_ERR_INVALID_ARG,
                "gfs_gms_inlier_mask: pair %d has an empty frame size", f);
    GFS_REQUIRE(p.n_matches == 0 || (p.kp1 && p.kp2 && p.query_idx && p.train_idx && inlier[f]), GFS_ERR_INVALID_ARG,
                "gfs_gms_inlier_mask: pair %d has NULL arrays", f);
    for (int i = 0; i < p.n_matches; i++)
      GFS_REQUIRE(p.query_idx[i] >= 0 && p.query_idx[i] < p.n1 && p.train_idx[i] >= 0 && p.train_idx[i] < p.n2, GFS_ERR_INVALID_ARG,
                  "gfs_gms_inlier_mask: pair %d match %d references an unknown key-point", f, i);
    h->h_pairs.p[f] = GmsPair{p.n1, p.n2, p.n_matches, p.width1, p.height1, p.width2, p.height2, 0};
    if (p.n1) memcpy(h->h_kp1.p + (size_t)f * S, p.kp1, (size_t)p.n1 * sizeof(gfs_keypoint));
    if (p.n2) memcpy(h->h_kp2.p + (size_t)f * S, p.kp2, (size_t)p.n2 * sizeof(gfs_keypoint));
    if (p.n_matches) {
      memcpy(h->h_q.p + (size_t)f * S, p.query_idx, (size_t)p.n_matches * 4);
      memcpy(h->h_t.p + (size_t)f * S, p.train_idx, (size_t)p.n_matches * 4);
    }
  }
  hipStream_t s = h->stream;
  const size_t K = (size_t)S * B;
  GFS_HIP(hipMemcpyAsync(h->d_pairs.p, h->h_pairs.p, B * sizeof(GmsPair), hipMemcpyHostToDevice, s));
  GFS_HIP(hipMemcpyAsync(h->d_kp1.p, h->h_kp1.p, K * sizeof(gfs_keypoint), hipMemcpyHostToDevice, s));
  GFS_HIP(hipMemcpyAsync(h->d_kp2.p, h->h_kp2.p, K * sizeof(gfs_keypoint), hipMemcpyHostToDevice, s));
  GFS_HIP(hipMemcpyAsync(h->d_q.p, h->h_q.p, K * 4, hipMemcpyHostToDevice, s));
  GFS_HIP(hipMemcpyAsync(h->d_t.p, h->h_t.p, K * 4, hipMemcpyHostToDevice, s));
  GFS_LAUNCH("k_gms", k_gms, dim3(4, B), dim3(kGmsThreads), 0, s, h->d_pairs.p, GmsPair{}, h->d_kp1.p, h->d_kp2.p, (const int*)nullptr,
             (const int*)nullptr, S, h->d_q.p, h->d_t.p, S, h->d_vote.p);
  GFS_LAUNCH("k_gms_merge", k_gms_merge, dim3(B), dim3(kGmsThreads), 0, s, h->d_pairs.p, GmsPair{}, (const int*)nullptr,
             (const int*)nullptr, S, (const uint8_t*)h->d_vote.p, h->d_mask.p, h->d_counts.p);
  GFS_HIP(hipMemcpyAsync(h->h_mask.p, h->d_mask.p, K, hipMemcpyDeviceToHost, s));
  GFS_HIP(hipMemcpyAsync(h->h_counts.p, h->d_counts.p, B * 4, hipMemcpyDeviceToHost, s));
  GFS_HIP(hipStreamSynchronize(s));
  for (int f = 0; f < B; f++) {
    if (problems[f].n_matches) memcpy(inlier[f], h->h_mask.p + (size_t)f * S, (size_t)problems[f].n_matches);
    n_inliers[f] = h->h_counts.p[f];
  }
  return GFS_OK;
}

int gfs_gms_inlier_mask_batch_device(gfs_gms* h, const void* dev_kps1, const void* dev_n1, const void* dev_kps2, const void* dev_n2,
                                     int B, int kp_stride, const void* dev_train_idx, int width, int height, void* dev_mask,
                                     void* dev_counts, void* stream) {
  GFS_REQUIRE(h && dev_kps1 && dev_n1 && dev_kps2 && dev_n2 && dev_train_idx && dev_mask && dev_counts && B > 0, GFS_ERR_INVALID_ARG,
              "gfs_gms_inlier_mask_batch_device: invalid argument");
  GFS_REQUIRE(B <= h->max_batch && kp_stride <= h->max_kps && kp_stride > 0, GFS_ERR_CAPACITY,
              "gfs_gms_inlier_mask_batch_device: batch %d / stride %d exceed the capacity (%d / %d)", B, kp_stride, h->max_batch, h->max_kps);
  GFS_REQUIRE(width > 0 && height > 0, GFS_ERR_INVALID_ARG, "gfs_gms_inlier_mask_batch_device: empty frame size");
  std::lock_guard<std::mutex> lk(h->mu);
  GFS_HIP(hipSetDevice(h->device));
  hipStream_t s = stream ? (hipStream_t)stream : h->stream;
  const GmsPair common{0, 0, 0, width, height, width, height, 1};
  GFS_LAUNCH("k_gms", k_gms, dim3(4, B), dim3(kGmsThreads), 0, s, (const GmsPair*)nullptr, common, (const gfs_keypoint*)dev_kps1,
             (const gfs_keypoint*)dev_kps2, (const int*)dev_n1, (const int*)dev_n2, kp_stride, (const int*)nullptr,
             (const int*)dev_train_idx, kp_stride, h->d_vote.p);
  GFS_LAUNCH("k_gms_merge", k_gms_merge, dim3(B), dim3(kGmsThreads), 0, s, (const GmsPair*)nullptr, common, (const int*)dev_n1,
             (const int*)dev_n2, kp_stride, (const uint8_t*)h->d_vote.p, (uint8_t*)dev_mask, (int*)dev_counts);
  if (!stream) GFS_HIP(hipStreamSynchronize(s));
  return GFS_OK;
}

}  // extern "C"
