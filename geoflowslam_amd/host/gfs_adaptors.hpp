// C++ host-side mirror of the reference's four seams on top of the C ABI (include/gfs_abi.h).
//
// The reference is C++17 (OpenCV / Eigen types in its signatures).  Neither library exists in this image, so
// this header offers TWO layers:
//   1. namespace gfs_host: the same classes with plain std:: containers (always compiled; used by the examples
//      and by anything that does not want OpenCV / Eigen);
//   2. namespace ORB_SLAM3 (guarded by GFS_WITH_OPENCV / GFS_WITH_EIGEN): drop-in classes with the reference's
//      EXACT signatures —
//        ORBextractor::operator()(cv::InputArray, cv::InputArray, std::vector<cv::KeyPoint>&, cv::OutputArray,
//                                 std::vector<int>&)                      include/ORBextractor.h:61-64
//        ORBmatcher::DescriptorDistance(const cv::Mat&, const cv::Mat&)   include/ORBmatcher.h:41
//        bf_match(d1, d2, std::vector<cv::DMatch>&)                       src/ORBmatcher.cc:755-756
//        RegistrationGICP::RegisterPointClouds(...)                       include/RegistrationGICP.h:25-28
//      INTEGRATION.md shows where they are swapped in.
// Error behaviour follows the reference: operator() returns -1 on an empty image, asserts CV_8UC1; GICP never throws.
// Anything the GPU library reports as an error is raised as std::runtime_error (there is no CPU fallback).
#pragma once
#include <algorithm>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/gfs_abi.h"

namespace gfs_host {

inline void check(int rc, const char* what) {
  if (rc < 0) throw std::runtime_error(std::string(what) + ": " + gfs_last_error());
}

// ORB_SLAM3::ORBextractor (reference include/ORBextractor.h:46-118)
class ORBextractor {
 public:
  ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST, int max_rows = 720,
               int max_cols = 1280, int device = 0)
      : nlevels_(nlevels) {
    gfs_orb_config c;
    gfs_orb_default_config(&c);
    c.nfeatures = nfeatures;
    c.scale_factor = scaleFactor;
    c.nlevels = nlevels;
    c.ini_th_fast = iniThFAST;
    c.min_th_fast = minThFAST;
    c.max_rows = max_rows;
    c.max_cols = max_cols;
    c.device = device;
    check(gfs_orb_create(&c, &h_), "gfs_orb_create");
    cap_ = gfs_orb_max_keypoints(h_);
  }
  ~ORBextractor() { gfs_orb_destroy(h_); }
  ORBextractor(const ORBextractor&) = delete;
  ORBextractor& operator=(const ORBextractor&) = delete;

  // operator()(image, mask, keypoints, descriptors, vLappingArea): returns monoIndex, or -1 for an empty image
  int operator()(const uint8_t* image, int rows, int cols, int stride, std::vector<gfs_keypoint>& keypoints,
                 std::vector<uint8_t>& descriptors, const std::vector<int>& vLappingArea) {
    keypoints.assign(cap_, gfs_keypoint{});
    descriptors.assign((size_t)cap_ * 32, 0);
    int n = 0;
    const int lap0 = vLappingArea.size() > 0 ? vLappingArea[0] : 0, lap1 = vLappingArea.size() > 1 ? vLappingArea[1] : 0;
    const int r = gfs_orb_extract(h_, image, rows, cols, stride, lap0, lap1, keypoints.data(), descriptors.data(), cap_, &n);
    if (r < -1) check(r + 100, "gfs_orb_extract");
    keypoints.resize(n);
    descriptors.resize((size_t)n * 32);
    return r;
  }
  int GetLevels() const { return nlevels_; }
  std::vector<float> GetScaleFactors() const { return table(0); }
  std::vector<float> GetInverseScaleFactors() const { return table(1); }
  std::vector<float> GetScaleSigmaSquares() const { return table(2); }
  std::vector<float> GetInverseScaleSigmaSquares() const { return table(3); }
  gfs_orb* handle() { return h_; }

 private:
  std::vector<float> table(int which) const {
    std::vector<float> t[4];
    for (auto& v : t) v.resize(nlevels_);
    gfs_orb_get_tables(h_, t[0].data(), t[1].data(), t[2].data(), t[3].data(), nullptr, nullptr);
    return t[which];
  }
  gfs_orb* h_ = nullptr;
  int nlevels_, cap_ = 0;
};

struct DMatch {  // cv::DMatch
  int queryIdx, trainIdx, imgIdx;
  float distance;
};

// the brute-force part of ORB_SLAM3::ORBmatcher (reference src/ORBmatcher.cc:744-778, 2536-2550)
class ORBmatcher {
 public:
  explicit ORBmatcher(int max_rows = 8192, int device = 0) { check(gfs_matcher_create(device, max_rows, max_rows, 1, &h_), "gfs_matcher_create"); }
  ~ORBmatcher() { gfs_matcher_destroy(h_); }
  static int DescriptorDistance(const uint8_t* a, const uint8_t* b) { return gfs_hamming256(a, b); }
  // cv::BFMatcher(cv::NORM_HAMMING).match(query, train, matches)
  void match(const uint8_t* query, int nq, const uint8_t* train, int nt, std::vector<DMatch>& matches) {
    std::vector<int32_t> idx(nq > 0 ? nq : 1), dist(nq > 0 ? nq : 1);
    const int n = gfs_bf_match_hamming(h_, query, nq, train, nt, idx.data(), dist.data());
    check(n, "gfs_bf_match_hamming");
    matches.resize(n);
    for (int i = 0; i < n; i++) matches[i] = DMatch{i, idx[i], 0, (float)dist[i]};
  }

 private:
  gfs_matcher* h_ = nullptr;
};

// RegistrationGICP (reference include/RegistrationGICP.h:19-31); result = small_gicp::RegistrationResult
class RegistrationGICP {
 public:
  explicit RegistrationGICP(int max_points = 65536, int device = 0) { check(gfs_gicp_create(device, max_points, 1, &h_), "gfs_gicp_create"); }
  ~RegistrationGICP() { gfs_gicp_destroy(h_); }
  // target / source: arrays of (x, y, z, w) floats like std::vector<Eigen::Vector4f>; init: column-major 4x4
  gfs_gicp_result RegisterPointClouds(const float* target_points, int nt, const float* source_points, int ns,
                                      const double init_T_target_source[16]) {
    gfs_gicp_config cfg;
    gfs_gicp_default_config(&cfg);  // threads 4, voxel 0.02, max-corr 0.1, GICP (src/RegistrationGICP.cc:9-15)
    gfs_gicp_result r;
    check(gfs_gicp_align(h_, target_points, nt, source_points, ns, init_T_target_source, &cfg, &r), "gfs_gicp_align");
    return r;
  }
  // Streaming form for Tracking::PredictStateICP: the target is the source cloud of the previous call on this object, kept
  // preprocessed in HBM (bit-identical to RegisterPointClouds(previous source, source_points)).
  gfs_gicp_result RegisterNext(const float* source_points, int ns, const double init_T_target_source[16]) {
    gfs_gicp_config cfg;
    gfs_gicp_default_config(&cfg);
    gfs_gicp_result r;
    check(gfs_gicp_align_next(h_, source_points, ns, init_T_target_source, &cfg, &r), "gfs_gicp_align_next");
    return r;
  }

 private:
  gfs_gicp* h_ = nullptr;
};

// numeric core of Optimizer::LocalBundleAdjustment (reference include/Optimizer.h:62-65)
class LocalBundleAdjuster {
 public:
  LocalBundleAdjuster(int max_poses = 64, int max_points = 16384, int max_edges = 262144, int device = 0) {
    check(gfs_lba_create(device, max_poses, max_points, max_edges, &h_), "gfs_lba_create");
  }
  ~LocalBundleAdjuster() { gfs_lba_destroy(h_); }
  // returns false when *pbStopFlag was already set (the reference returns early, src/Optimizer.cc:1955-1956)
  bool solve(const gfs_lba_problem& p, gfs_lba_solution& s, const bool* pbStopFlag) {
    volatile int stop = (pbStopFlag && *pbStopFlag) ? 1 : 0;
    const int rc = gfs_lba_solve(h_, &p, &s, pbStopFlag ? &stop : nullptr);
    if (rc == GFS_ERR_STOPPED) return false;
    check(rc, "gfs_lba_solve");
    return true;
  }

 private:
  gfs_lba* h_ = nullptr;
};

// gms_matcher(kp1, size1, kp2, size2, matches).GetInlierMask(mask, false, false) (reference Thirdparty/GMS/include/gms_matcher.h;
// the filter SearchWithGMS applies to the brute-force matches, src/ORBmatcher.cc:761-762)
class GmsMatcher {
 public:
  explicit GmsMatcher(int max_keypoints = 8192, int device = 0) { check(gfs_gms_create(device, max_keypoints, 1, &h_), "gfs_gms_create"); }
  ~GmsMatcher() { gfs_gms_destroy(h_); }
  // matches[i] = (queryIdx, trainIdx); returns the number of inliers, mask[i] = vbInliers[i]
  int GetInlierMask(const gfs_keypoint* kp1, int n1, int w1, int h1, const gfs_keypoint* kp2, int n2, int w2, int h2,
                    const std::vector<int32_t>& query_idx, const std::vector<int32_t>& train_idx, std::vector<uint8_t>& mask) {
    const int n = (int)query_idx.size();
    mask.assign((size_t)std::max(n, 1), 0);
    gfs_gms_problem p{n1, n2, kp1, kp2, w1, h1, w2, h2, n, query_idx.data(), train_idx.data()};
    uint8_t* mp = mask.data();
    int32_t nin = 0;
    check(gfs_gms_inlier_mask(h_, &p, 1, &mp, &nin), "gfs_gms_inlier_mask");
    mask.resize((size_t)n);
    return nin;
  }

 private:
  gfs_gms* h_ = nullptr;
};

// The optical-flow front end: cv::buildOpticalFlowPyramid (reference src/Frame.cc:373), ORBmatcher::fbKltTracking
// (src/ORBmatcher.cc:2186-2297 == Tracking::fbKltTracking, src/Tracking.cc:3262-3366).  A Pyramid is what Frame::mImGray holds in
// the reference (a std::vector<cv::Mat>), kept in HBM; build it once per frame, use it as `cur` and then as `prev`.
class KltTracker {
 public:
  class Pyramid {
   public:
    explicit Pyramid(KltTracker& t) : t_(t) { check(gfs_klt_pyramid_create(t.h_, &p_), "gfs_klt_pyramid_create"); }
    ~Pyramid() { gfs_klt_pyramid_destroy(p_); }
    Pyramid(const Pyramid&) = delete;
    Pyramid& operator=(const Pyramid&) = delete;
    // cv::buildOpticalFlowPyramid(image, pyr, Size(win, win), max_level)
    void build(const uint8_t* image, int stride) { check(gfs_klt_build_pyramid(t_.h_, p_, &image, stride, 1), "gfs_klt_build_pyramid"); }
    gfs_klt_pyramid* get() const { return p_; }

   private:
    KltTracker& t_;
    gfs_klt_pyramid* p_ = nullptr;
  };

  KltTracker(int width, int height, int nwinsize, int max_level = 3, int max_points = 8192, int device = 0) {
    check(gfs_klt_create(device, width, height, nwinsize, max_level, 1, max_points, &h_), "gfs_klt_create");
  }
  ~KltTracker() { gfs_klt_destroy(h_); }
  gfs_klt* handle() const { return h_; }
  // fbKltTracking(vprevpyr, vcurpyr, nwinsize, nbpyrlvl, ferr, fmax_fbklt_dist, vkps, vpriorkps, vkpstatus): points are
  // (x, y) float pairs as cv::Point2f; vpriorkps is updated in place, vkpstatus is resized to vkps.size() / 2.
  void fbKltTracking(const Pyramid& vprevpyr, const Pyramid& vcurpyr, int nbpyrlvl, float ferr, float fmax_fbklt_dist,
                     const std::vector<float>& vkps, std::vector<float>& vpriorkps, std::vector<uint8_t>& vkpstatus) {
    const int32_t n = (int32_t)(vkps.size() / 2);
    vkpstatus.assign((size_t)std::max(n, 1), 0);
    if (n == 0) {  // src/ORBmatcher.cc:2197-2199: nothing is touched
      vkpstatus.clear();
      return;
    }
    const float* kp = vkps.data();
    float* pr = vpriorkps.data();
    uint8_t* st = vkpstatus.data();
    int32_t good = 0;
    check(gfs_klt_fb_track(h_, vprevpyr.get(), vcurpyr.get(), 1, &n, &kp, &pr, &st, &good, nbpyrlvl, ferr, fmax_fbklt_dist), "gfs_klt_fb_track");
    vkpstatus.resize((size_t)n);
  }

 private:
  gfs_klt* h_ = nullptr;
};

// cv::findFundamentalMat(points1, points2, cv::FM_RANSAC, threshold, confidence, status), 8 or more points (reference call
// sites src/ORBmatcher.cc:236, 2399, 2463; src/Tracking.cc:1974)
class FundamentalMatcher {
 public:
  explicit FundamentalMatcher(int max_points = 8192, int device = 0) { check(gfs_fmat_create(device, max_points, 1, &h_), "gfs_fmat_create"); }
  ~FundamentalMatcher() { gfs_fmat_destroy(h_); }
  // points: (x, y) float pairs as cv::Point2f; returns the consensus size, status[i] = 0 / 1, F = row-major 3x3 (zeros: no model)
  int findFundamentalMat(const std::vector<float>& points1, const std::vector<float>& points2, double threshold, double confidence,
                         std::vector<uint8_t>& status, double F[9] = nullptr) {
    const int32_t n = (int32_t)(points1.size() / 2);
    status.assign((size_t)std::max(n, 1), 0);
    const float* a = points1.data();
    const float* b = points2.data();
    uint8_t* st = status.data();
    int32_t n_in = 0;
    check(gfs_find_fundamental_ransac(h_, 1, &n, &a, &b, threshold, confidence, 1000, &st, F, &n_in), "gfs_find_fundamental_ransac");
    status.resize((size_t)n);
    return n_in;
  }

 private:
  gfs_fmat* h_ = nullptr;
};

// ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono) on flattened frames (reference src/ORBmatcher.cc:1853-2063;
// see INTEGRATION.md §6 for the flattening of Frame / MapPoint)
class ProjectionMatcher {
 public:
  ProjectionMatcher(int max_last = 8192, int max_cur = 4096, int device = 0) { check(gfs_sbp_create(device, max_last, max_cur, 1, &h_), "gfs_sbp_create"); }
  ~ProjectionMatcher() { gfs_sbp_destroy(h_); }
  // cur_match: p.n_cur entries (>= 0 new map point = last-list entry, -1 untouched, -2 reset to NULL); returns nmatches
  int SearchByProjection(const gfs_sbp_problem& p, std::vector<int32_t>& cur_match) {
    cur_match.assign((size_t)std::max(p.n_cur, 1), -1);
    int32_t* ptr = cur_match.data();
    int32_t n = 0;
    check(gfs_search_by_projection(h_, &p, 1, &ptr, &n), "gfs_search_by_projection");
    cur_match.resize((size_t)p.n_cur);
    return n;
  }

 private:
  gfs_sbp* h_ = nullptr;
};

// Optimizer::PoseOptimization on a flattened frame (reference src/Optimizer.cc:763-1098)
class PoseOptimizer {
 public:
  PoseOptimizer(int max_obs = 8192, int device = 0) { check(gfs_pose_create(device, max_obs, 1, &h_), "gfs_pose_create"); }
  ~PoseOptimizer() { gfs_pose_destroy(h_); }
  // returns nInitialCorrespondences - nBad; s.outlier / s.chi2 must point to p.n_obs entries
  int PoseOptimization(const gfs_pose_problem& p, gfs_pose_solution& s) {
    check(gfs_pose_optimize(h_, &p, 1, &s), "gfs_pose_optimize");
    return s.n_inliers;
  }

 private:
  gfs_pose* h_ = nullptr;
};

}  // namespace gfs_host

#if defined(GFS_WITH_OPENCV)
#include <opencv2/core/core.hpp>
#include <opencv2/features2d/features2d.hpp>
#include <cassert>
namespace ORB_SLAM3 {
// Drop-in for the reference class (same name, same virtual operator()): see INTEGRATION.md §1.
class ORBextractor {
 public:
  enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };
  ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST)
      : impl_(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST) {}
  virtual ~ORBextractor() {}
  virtual int operator()(cv::InputArray _image, cv::InputArray /*_mask*/, std::vector<cv::KeyPoint>& _keypoints,
                         cv::OutputArray _descriptors, std::vector<int>& vLappingArea) {
    if (_image.empty()) return -1;
    cv::Mat image = _image.getMat();
    assert(image.type() == CV_8UC1);
    std::vector<gfs_keypoint> k;
    std::vector<uint8_t> d;
    const int mono = impl_(image.data, image.rows, image.cols, (int)image.step, k, d, vLappingArea);
    static_assert(sizeof(cv::KeyPoint) == sizeof(gfs_keypoint), "cv::KeyPoint layout");
    _keypoints.resize(k.size());
    if (!k.empty()) memcpy((void*)_keypoints.data(), k.data(), k.size() * sizeof(gfs_keypoint));
    if (k.empty())
      _descriptors.release();
    else {
      _descriptors.create((int)k.size(), 32, CV_8U);
      memcpy(_descriptors.getMat().data, d.data(), d.size());
    }
    return mono;
  }
  int GetLevels() { return impl_.GetLevels(); }
  std::vector<float> GetScaleFactors() { return impl_.GetScaleFactors(); }
  std::vector<float> GetInverseScaleFactors() { return impl_.GetInverseScaleFactors(); }
  std::vector<float> GetScaleSigmaSquares() { return impl_.GetScaleSigmaSquares(); }
  std::vector<float> GetInverseScaleSigmaSquares() { return impl_.GetInverseScaleSigmaSquares(); }

 private:
  gfs_host::ORBextractor impl_;
};
}  // namespace ORB_SLAM3
#endif
