// The reference-side binding code of INTEGRATION.md as a header a GeoFlow-SLAM maintainer adds to the reference tree: the four seams
// of SURVEY.md 8(b) written against the reference's OWN types (cv::Mat / cv::KeyPoint / cv::DMatch, Eigen, Sophus,
// ORB_SLAM3::ORBextractor, small_gicp::RegistrationResult) on top of the C ABI (include/gfs_abi.h).  Include it AFTER the reference
// headers it names (include/ORBextractor.h, small_gicp/registration/registration_result.hpp); link with -lgfs_hip.
//
// OpenCV / Eigen / Sophus are not in this repository's image, so this header cannot be linked here; it IS compiled:
// tests/test_host_logic.py::test_reference_dropins_compile runs g++ -fsyntax-only over it against the reference's real
// ORBextractor.h / registration_result.hpp (when /root/reference is present) and declaration-only stand-ins for the three libraries
// (tests/host/stubs/), so the signatures, overrides and struct layouts the text of INTEGRATION.md promises are checked by a compiler.
#pragma once
#include <cassert>
#include <cstring>
#include <numeric>
#include <stdexcept>
#include <vector>

#include "../../include/gfs_abi.h"

namespace gfs_dropin {

// ---- 1. ORBextractor: a subclass selected by ORBextractor::make_extractor (src/ORBextractor.cc:1253-1265) ----------------------
//      include/ORBextractor.h:47   enum EXTRACTOR_TYPE { ORB = 0, SUPERPOINT, ORB_MI355X };
//      src/ORBextractor.cc:1253    case EXTRACTOR_TYPE::ORB_MI355X: return new gfs_dropin::GfsORBextractor(nfeatures, scaleFactor, ...);
class GfsORBextractor : public ORB_SLAM3::ORBextractor {  // keeps the getters and mvImagePyramid of the base
 public:
  // mvImagePyramid (include/ORBextractor.h:82, a public member) is read by the stereo matcher only (src/Frame.cc:1159-1256); an
  // RGB-D / monocular frame never looks at it.  Set this to have operator() copy the un-blurred levels back after every call.
  bool fill_image_pyramid = false;
  GfsORBextractor(int nf, float sf, int nl, int ini, int mn, int max_rows = 1080, int max_cols = 1920) : ORBextractor(nf, sf, nl, ini, mn) {
    gfs_orb_config c;
    gfs_orb_default_config(&c);
    c.nfeatures = nf;
    c.scale_factor = sf;
    c.nlevels = nl;
    c.ini_th_fast = ini;
    c.min_th_fast = mn;
    c.max_rows = max_rows;
    c.max_cols = max_cols;
    if (gfs_orb_create(&c, &h_) != GFS_OK) throw std::runtime_error(gfs_last_error());
    cap_ = gfs_orb_max_keypoints(h_);
  }
  ~GfsORBextractor() override { gfs_orb_destroy(h_); }
  int operator()(cv::InputArray _image, cv::InputArray /*_mask*/, std::vector<cv::KeyPoint>& _keypoints, cv::OutputArray _descriptors,
                 std::vector<int>& vLappingArea) override {
    if (_image.empty()) return -1;  // src/ORBextractor.cc:1150
    cv::Mat image = _image.getMat();
    assert(image.type() == CV_8UC1);  // :1153
    static_assert(sizeof(cv::KeyPoint) == sizeof(gfs_keypoint), "cv::KeyPoint and gfs_keypoint must have one layout (28 bytes)");
    _keypoints.resize((size_t)cap_);
    cv::Mat desc(cap_, 32, CV_8U);
    int n = 0;
    const int mono = gfs_orb_extract(h_, image.data, image.rows, image.cols, (int)(size_t)image.step, vLappingArea[0], vLappingArea[1],
                                     reinterpret_cast<gfs_keypoint*>(_keypoints.data()), desc.data, cap_, &n);
    if (GFS_ORB_EXTRACT_FAILED(mono)) throw std::runtime_error(gfs_last_error());
    _keypoints.resize((size_t)n);
    if (n == 0)
      _descriptors.release();
    else
      desc.rowRange(0, n).copyTo(_descriptors);  // :1168-1171
    if (fill_image_pyramid) {  // ComputePyramid leaves the resized levels here (src/ORBextractor.cc:1227-1251); the blur works on clones
      mvImagePyramid.resize((size_t)GetLevels());
      for (int level = 0; level < GetLevels(); ++level) {
        int rows = 0, cols = 0;
        if (gfs_orb_level_size(h_, level, &rows, &cols) != GFS_OK) throw std::runtime_error(gfs_last_error());
        mvImagePyramid[(size_t)level].create(rows, cols, CV_8UC1);  // continuous rows * cols bytes
        if (gfs_orb_fetch_level(h_, 0, level, 0, mvImagePyramid[(size_t)level].data) != GFS_OK) throw std::runtime_error(gfs_last_error());
      }
    }
    return mono;
  }

 private:
  gfs_orb* h_ = nullptr;
  int cap_ = 0;
};

// ---- 2. ORBmatcher's brute-force sites (src/ORBmatcher.cc:755-756, 805-806, 888-889) and the GMS filter after them (:761-762) -----
//      BFMatcher matcher(NORM_HAMMING); matcher.match(d1, d2, matches_all);   ->   gfs_dropin::bf_match(d1, d2, matches_all);
inline void bf_match(const cv::Mat& d1, const cv::Mat& d2, std::vector<cv::DMatch>& matches_all, std::vector<int32_t>* train_idx = nullptr) {
  static thread_local gfs_matcher* m = [] {
    gfs_matcher* x = nullptr;
    if (gfs_matcher_create(0, 8192, 8192, 1, &x) != GFS_OK) throw std::runtime_error(gfs_last_error());
    return x;
  }();
  std::vector<int32_t> idx((size_t)d1.rows), dist((size_t)d1.rows);
  const int n = gfs_bf_match_hamming(m, d1.data, d1.rows, d2.data, d2.rows, idx.data(), dist.data());
  if (n < 0) throw std::runtime_error(gfs_last_error());
  matches_all.resize((size_t)n);
  for (int i = 0; i < n; ++i) matches_all[(size_t)i] = cv::DMatch(i, idx[(size_t)i], 0, (float)dist[(size_t)i]);
  if (train_idx) train_idx->assign(idx.begin(), idx.begin() + n);
}
//      gms_matcher gms(kp1, frameSize, kp2, frameSize, matches_all); nmatches = gms.GetInlierMask(vbInliers, false, false);
inline int gms_inlier_mask(const std::vector<cv::KeyPoint>& kp1, const std::vector<cv::KeyPoint>& kp2, const cv::Size& frameSize,
                           const std::vector<cv::DMatch>& matches_all, std::vector<bool>& vbInliers) {
  static thread_local gfs_gms* g = [] {
    gfs_gms* x = nullptr;
    if (gfs_gms_create(0, 8192, 1, &x) != GFS_OK) throw std::runtime_error(gfs_last_error());
    return x;
  }();
  const int n = (int)matches_all.size();
  std::vector<int32_t> q((size_t)n), t((size_t)n);
  for (int i = 0; i < n; ++i) {
    q[(size_t)i] = matches_all[(size_t)i].queryIdx;
    t[(size_t)i] = matches_all[(size_t)i].trainIdx;
  }
  gfs_gms_problem gp{(int32_t)kp1.size(), (int32_t)kp2.size(), reinterpret_cast<const gfs_keypoint*>(kp1.data()),
                     reinterpret_cast<const gfs_keypoint*>(kp2.data()), frameSize.width, frameSize.height, frameSize.width, frameSize.height,
                     n, q.data(), t.data()};
  std::vector<uint8_t> mask((size_t)n);
  uint8_t* mp = mask.data();
  int32_t nin = 0;
  if (gfs_gms_inlier_mask(g, &gp, 1, &mp, &nin) != GFS_OK) throw std::runtime_error(gfs_last_error());
  vbInliers.assign(mask.begin(), mask.end());
  return nin;
}

// ---- 3. RegistrationGICP::RegisterPointClouds (include/RegistrationGICP.h:25-28, src/RegistrationGICP.cc:5-20): the body ------------
inline small_gicp::RegistrationResult RegisterPointClouds(const std::vector<Eigen::Vector4f>& target_points,
                                                          const std::vector<Eigen::Vector4f>& source_points,
                                                          const Eigen::Isometry3d& init_T_target_source) {
  static_assert(sizeof(Eigen::Vector4f) == 16, "Vector4f is 16 contiguous bytes (x, y, z, w): vector<Vector4f> is the [N][4] float array of the ABI");
  static thread_local gfs_gicp* h = [] {
    gfs_gicp* g = nullptr;
    if (gfs_gicp_create(0, 131072, 1, &g) != GFS_OK) throw std::runtime_error(gfs_last_error());
    return g;
  }();
  gfs_gicp_config cfg;
  gfs_gicp_default_config(&cfg);  // threads 4 / voxel 0.02 / max-corr 0.1 / GICP (src/RegistrationGICP.cc:9-15)
  gfs_gicp_result r;
  if (gfs_gicp_align(h, target_points[0].data(), (int)target_points.size(), source_points[0].data(), (int)source_points.size(),
                     init_T_target_source.matrix().data(), &cfg, &r) != GFS_OK)
    throw std::runtime_error(gfs_last_error());
  small_gicp::RegistrationResult out(Eigen::Isometry3d(Eigen::Map<Eigen::Matrix4d>(r.T_target_source)));
  out.converged = r.converged != 0;
  out.iterations = (size_t)r.iterations;
  out.num_inliers = (size_t)r.num_inliers;
  out.H = Eigen::Map<Eigen::Matrix<double, 6, 6>>(r.H);
  out.b = Eigen::Map<Eigen::Matrix<double, 6, 1>>(r.b);
  out.error = r.error;
  return out;  // the gate at src/Tracking.cc:3394 is unchanged
}

// ---- 4. Optimizer::LocalBundleAdjustment: the Access policy of gfs_host::LocalBundleAdjustment (gfs_adaptors.hpp) for the
//         reference's KeyFrame / MapPoint (Sophus::SE3f, Eigen::Vector3f <-> plain floats; coeffs() is x, y, z, w) ------------------
template <class KeyFrame, class MapPoint>
struct LbaAccess {
  static void pose(const KeyFrame* k, float q[4], float t[3]) {
    const Sophus::SE3f Tcw = const_cast<KeyFrame*>(k)->GetPose();
    std::memcpy(q, Tcw.unit_quaternion().coeffs().data(), 16);
    std::memcpy(t, Tcw.translation().data(), 12);
  }
  static void set_pose(KeyFrame* k, const float q[4], const float t[3]) {
    k->SetPose(Sophus::SE3f(Eigen::Quaternionf(q[3], q[0], q[1], q[2]), Eigen::Vector3f(t[0], t[1], t[2])));
  }
  static void world_pos(const MapPoint* p, float x[3]) {
    const Eigen::Vector3f w = const_cast<MapPoint*>(p)->GetWorldPos();
    std::memcpy(x, w.data(), 12);
  }
  static void set_world_pos(MapPoint* p, const float x[3]) { p->SetWorldPos(Eigen::Vector3f(x[0], x[1], x[2])); }
};

}  // namespace gfs_dropin
