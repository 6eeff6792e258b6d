// C++ host-side mirror of the reference's four seams on top of the C ABI (include/gfs_abi.h).
//
// The reference is C++17 (OpenCV / Eigen types in its signatures).  Neither library exists in this image, so there are TWO layers:
//   1. this header, namespace gfs_host: the same classes with plain std:: containers (always compiled; used by the examples, the
//      tests and by anything that does not want OpenCV / Eigen);
//   2. gfs_reference_dropins.hpp beside it, namespace gfs_dropin: the code a maintainer adds to the reference tree, written against
//      the reference's own types --
//        GfsORBextractor : ORB_SLAM3::ORBextractor, operator()(cv::InputArray, cv::InputArray, std::vector<cv::KeyPoint>&,
//                                                              cv::OutputArray, std::vector<int>&)   include/ORBextractor.h:61-64
//        bf_match(d1, d2, std::vector<cv::DMatch>&), gms_inlier_mask(...)                            src/ORBmatcher.cc:755-762
//        RegisterPointClouds(...) -> small_gicp::RegistrationResult                                  include/RegistrationGICP.h:25-28
//        LbaAccess<KeyFrame, MapPoint> for gfs_host::LocalBundleAdjustment                           src/Optimizer.cc:1588-2040
//      compile-checked against the reference's real headers over declaration-only OpenCV / Eigen / Sophus stand-ins
//      (tests/test_host_logic.py::test_reference_dropins_compile).  INTEGRATION.md shows where they are swapped in.
// Error behaviour follows the reference: operator() returns -1 on an empty image, asserts CV_8UC1; GICP never throws.
// Anything the GPU library reports as an error is raised as std::runtime_error (there is no CPU fallback).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <list>
#include <map>
#include <mutex>
#include <set>
#include <stdexcept>
#include <string>
#include <tuple>
#include <type_traits>
#include <utility>
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
    // the caller's flag itself goes down (a C++ bool is one byte): the solver reads it live, at the top of every iteration and
    // after every trial step, so an mbAbortBA raised by the tracking thread WHILE the adjustment runs ends it (setForceStopFlag)
    static_assert(sizeof(bool) == 1, "gfs_lba_solve_bool reads the flag as one byte");
    const int rc = gfs_lba_solve_bool(h_, &p, &s, reinterpret_cast<const volatile unsigned char*>(pbStopFlag));
    if (rc == GFS_ERR_STOPPED) return false;
    check(rc, "gfs_lba_solve");
    return true;
  }
  // Optimizer::LocalBundleAdjustment on the reference's own KeyFrame / MapPoint / Map classes (see LocalBundleAdjustment below)
  template <class Access, class KeyFrame, class Map>
  void LocalBundleAdjustment(KeyFrame* pKF, bool* pbStopFlag, Map* pMap, int& num_fixedKF, int& num_OptKF, int& num_MPs,
                             int& num_edges);

 private:
  gfs_lba* h_ = nullptr;
};

// ------------------------------------------------------------------------------------------------------------------------
// Optimizer::LocalBundleAdjustment(KeyFrame* pKF, bool* pbStopFlag, bool pbICPFlag, Map* pMap, int& num_fixedKF,
//                                  int& num_OptKF, int& num_MPs, int& num_edges)          reference src/Optimizer.cc:1588-2040
// as real code around the flat solver: the pointer-graph gather (:1592-1660), the vertices and edges of the g2o graph flattened
// into a gfs_lba_problem in the reference's creation order (:1686-1721, 1816-1952), the stop-flag checks (:1679, 1955-1956),
// the chi2 / depth classification (:1961-1999) and the write-back under mMutexMapUpdate (:2003-2039).
//
// It is a template over the reference's own classes: KeyFrame, MapPoint and Map are used through exactly the members the
// reference function uses (mnId, mnBALocalForKF, mnBAFixedForKF, isBad(), GetMap(), GetVectorCovisibleKeyFrames(),
// GetMapPointMatches(), GetObservations(), mvKeysUn[i].pt / .octave, mvuRight, mvInvLevelSigma2, fx, fy, cx, cy, mbf, mpCamera2,
// EraseMapPointMatch(), EraseObservation(), UpdateNormalAndDepth(), GetInitKFid(), mMutexMapUpdate, IncreaseChangeIndex(),
// msOptKFs / msFixedKFs).  The four places that touch Sophus / Eigen value types go through `Access`:
//     static void pose(const KeyFrame*, float q_xyzw[4], float t[3]);        // GetPose(): unit_quaternion(), translation()
//     static void set_pose(KeyFrame*, const float q_xyzw[4], const float t[3]);   // SetPose(Sophus::SE3f(q, t))
//     static void world_pos(const MapPoint*, float p[3]);                    // GetWorldPos()
//     static void set_world_pos(MapPoint*, const float p[3]);                // SetWorldPos()
// (INTEGRATION.md section 4 gives the Access for the reference's types; tests/host/lba_adaptor_test.cpp one for plain structs.)
// `solve(problem, solution, stop)` is the numeric core: LocalBundleAdjuster::solve below (gfs_lba_solve on the GPU).
//
// Arithmetic the reference performs on the way in and out, reproduced here:
//   * poses: Sophus::SE3<float> -> g2o::SE3Quat(q.cast<double>(), t.cast<double>()) (the constructor's normalizeRotation() is
//     part of gfs_lba_solve), back through .cast<float>();
//   * points: Eigen::Vector3f -> cast<double>() and back;
//   * observations: kpUn.pt.x, kpUn.pt.y, mvuRight[idx] are floats assigned to doubles;
//   * information: Identity * invSigma2 with `const float& invSigma2` -> the float value as a double;
//   * Huber deltas: `const float thHuberMono = sqrt(5.991)` -> (double)(float)sqrt(5.991), same for sqrt(7.815).
// Not supported (GFS_ERR_UNSUPPORTED is raised): key-frames with a second camera (mpCamera2, EdgeSE3ProjectXYZToBody
// :1906-1949).  The ICP block :1762-1814 is dead code in the reference and has no counterpart.
// ------------------------------------------------------------------------------------------------------------------------
struct LbaFlat {  // the flattened graph, in the reference's vertex / edge creation order
  std::vector<double> pose_q, pose_t, points, edge_obs, edge_inv_sigma2;
  std::vector<uint8_t> pose_fixed, edge_stereo;
  std::vector<int32_t> edge_pose, edge_point;
  gfs_lba_problem problem{};
  void finish(double fx, double fy, double cx, double cy, double bf) {
    problem.n_poses = (int32_t)pose_fixed.size();
    problem.n_points = (int32_t)(points.size() / 3);
    problem.n_edges = (int32_t)edge_pose.size();
    problem.pose_q = pose_q.data();
    problem.pose_t = pose_t.data();
    problem.pose_fixed = pose_fixed.data();
    problem.points = points.data();
    problem.edge_pose = edge_pose.data();
    problem.edge_point = edge_point.data();
    problem.edge_obs = edge_obs.data();
    problem.edge_inv_sigma2 = edge_inv_sigma2.data();
    problem.edge_stereo = edge_stereo.data();
    problem.fx = fx;
    problem.fy = fy;
    problem.cx = cx;
    problem.cy = cy;
    problem.bf = bf;
    const float thHuberMono = (float)std::sqrt(5.991), thHuberStereo = (float)std::sqrt(7.815);  // src/Optimizer.cc:1728-1729
    problem.huber_mono = thHuberMono;
    problem.huber_stereo = thHuberStereo;
    problem.iterations = 10;  // optimizer.optimize(10), :1959
  }
};

template <class Access, class KeyFrame, class MapPoint, class Map, class Solve>
void LocalBundleAdjustment(Solve&& solve, KeyFrame* pKF, bool* pbStopFlag, Map* pMap, int& num_fixedKF, int& num_OptKF,
                           int& /*num_MPs: never written by the reference either*/, int& num_edges) {
  // ---- Local KeyFrames: first breadth search from the current key-frame (:1592-1607)
  std::list<KeyFrame*> lLocalKeyFrames;
  lLocalKeyFrames.push_back(pKF);
  pKF->mnBALocalForKF = pKF->mnId;
  Map* pCurrentMap = pKF->GetMap();
  const std::vector<KeyFrame*> vNeighKFs = pKF->GetVectorCovisibleKeyFrames();
  for (int i = 0, iend = (int)vNeighKFs.size(); i < iend; i++) {
    KeyFrame* pKFi = vNeighKFs[i];
    pKFi->mnBALocalForKF = pKF->mnId;
    if (!pKFi->isBad() && pKFi->GetMap() == pCurrentMap) lLocalKeyFrames.push_back(pKFi);
  }
  // ---- Local MapPoints seen in local key-frames (:1609-1634)
  num_fixedKF = 0;
  std::list<MapPoint*> lLocalMapPoints;
  for (KeyFrame* pKFi : lLocalKeyFrames) {
    if (pKFi->mnId == pMap->GetInitKFid()) num_fixedKF = 1;
    std::vector<MapPoint*> vpMPs = pKFi->GetMapPointMatches();
    for (MapPoint* pMP : vpMPs)
      if (pMP)
        if (!pMP->isBad() && pMP->GetMap() == pCurrentMap)
          if (pMP->mnBALocalForKF != pKF->mnId) {
            lLocalMapPoints.push_back(pMP);
            pMP->mnBALocalForKF = pKF->mnId;
          }
  }
  // ---- Fixed key-frames: see local MapPoints but are not local (:1636-1660)
  std::list<KeyFrame*> lFixedCameras;
  for (MapPoint* pMP : lLocalMapPoints) {
    auto observations = pMP->GetObservations();  // std::map<KeyFrame*, std::tuple<int, int>>
    for (auto mit = observations.begin(), mend = observations.end(); mit != mend; ++mit) {
      KeyFrame* pKFi = mit->first;
      if (pKFi->mnBALocalForKF != pKF->mnId && pKFi->mnBAFixedForKF != pKF->mnId) {
        pKFi->mnBAFixedForKF = pKF->mnId;
        if (!pKFi->isBad() && pKFi->GetMap() == pCurrentMap) lFixedCameras.push_back(pKFi);
      }
    }
  }
  num_fixedKF = (int)lFixedCameras.size() + num_fixedKF;
  if (num_fixedKF == 0) return;  // "LM-LBA: There are 0 fixed KF in the optimizations, LBA aborted" (:1662-1667)

  // ---- vertices (:1686-1721): pose index = creation order, local key-frames first
  LbaFlat F;
  std::map<const KeyFrame*, int32_t> pose_index;  // == optimizer.vertex(pKFi->mnId) != NULL
  pCurrentMap->msOptKFs.clear();
  pCurrentMap->msFixedKFs.clear();
  auto add_pose = [&](KeyFrame* pKFi, bool fixed) {
    float q[4], t[3];
    Access::pose(pKFi, q, t);
    pose_index[pKFi] = (int32_t)F.pose_fixed.size();
    for (int k = 0; k < 4; k++) F.pose_q.push_back((double)q[k]);  // .cast<double>()
    for (int k = 0; k < 3; k++) F.pose_t.push_back((double)t[k]);
    F.pose_fixed.push_back(fixed ? 1 : 0);
  };
  for (KeyFrame* pKFi : lLocalKeyFrames) {
    add_pose(pKFi, pKFi->mnId == pMap->GetInitKFid());  // vSE3->setFixed(pKFi->mnId == pMap->GetInitKFid())
    pCurrentMap->msOptKFs.insert(pKFi->mnId);
  }
  num_OptKF = (int)lLocalKeyFrames.size();
  for (KeyFrame* pKFi : lFixedCameras) {
    add_pose(pKFi, true);
    pCurrentMap->msFixedKFs.insert(pKFi->mnId);
  }
  // ---- MapPoint vertices and edges (:1816-1952).  The mono, body and stereo edges live in three vectors in the reference and
  //      are classified in that order afterwards; here every edge records its kind and (key-frame, point).
  struct EdgeRef {
    KeyFrame* kf;
    MapPoint* mp;
  };
  std::vector<EdgeRef> vpEdgesMono, vpEdgesStereo;
  std::vector<int32_t> mono_edge, stereo_edge;  // flat edge index of the k-th mono / stereo edge
  int nEdges = 0;
  int32_t point_index = 0;
  for (MapPoint* pMP : lLocalMapPoints) {
    float X[3];
    Access::world_pos(pMP, X);
    for (int k = 0; k < 3; k++) F.points.push_back((double)X[k]);
    const auto observations = pMP->GetObservations();
    for (auto mit = observations.begin(), mend = observations.end(); mit != mend; ++mit) {
      KeyFrame* pKFi = mit->first;
      if (!pKFi->isBad() && pKFi->GetMap() == pCurrentMap) {
        const int leftIndex = std::get<0>(mit->second);
        const auto pit = pose_index.find(pKFi);
        if (leftIndex != -1) {
          const bool stereo = pKFi->mvuRight[leftIndex] >= 0;
          if (pit != pose_index.end()) {  // optimizer.vertex(pKFi->mnId) == NULL -> continue
            const auto& kpUn = pKFi->mvKeysUn[leftIndex];
            F.edge_pose.push_back(pit->second);
            F.edge_point.push_back(point_index);
            F.edge_obs.push_back((double)kpUn.pt.x);
            F.edge_obs.push_back((double)kpUn.pt.y);
            F.edge_obs.push_back(stereo ? (double)pKFi->mvuRight[leftIndex] : 0.0);
            const float& invSigma2 = pKFi->mvInvLevelSigma2[kpUn.octave];
            F.edge_inv_sigma2.push_back((double)invSigma2);
            F.edge_stereo.push_back(stereo ? 1 : 0);
            (stereo ? vpEdgesStereo : vpEdgesMono).push_back(EdgeRef{pKFi, pMP});
            (stereo ? stereo_edge : mono_edge).push_back((int32_t)F.edge_pose.size() - 1);
            nEdges++;
          } else {
            continue;  // (the reference's `continue` also skips the second-camera block of this observation)
          }
        }
        if (pKFi->mpCamera2 && std::get<1>(mit->second) != -1)
          throw std::runtime_error("LocalBundleAdjustment: second-camera observations (EdgeSE3ProjectXYZToBody) are not supported");
      }
    }
    point_index++;
  }
  num_edges = nEdges;
  if (pbStopFlag)
    if (*pbStopFlag) return;  // :1955-1956
  F.finish(pKF->fx, pKF->fy, pKF->cx, pKF->cy, pKF->mbf);

  // ---- optimizer.initializeOptimization(); optimizer.optimize(10);
  std::vector<double> out_q(F.pose_q.size()), out_t(F.pose_t.size()), out_p(F.points.size()), chi2((size_t)std::max(nEdges, 1));
  std::vector<uint8_t> depth_pos((size_t)std::max(nEdges, 1));
  gfs_lba_solution S{};
  S.pose_q = out_q.data();
  S.pose_t = out_t.data();
  S.points = out_p.data();
  S.edge_chi2 = chi2.data();
  S.edge_depth_positive = depth_pos.data();
  if (!solve(F.problem, S, pbStopFlag)) return;

  // ---- check inlier observations (:1961-1999): mono edges first, then stereo, each in creation order
  std::vector<std::pair<KeyFrame*, MapPoint*>> vToErase;
  vToErase.reserve(vpEdgesMono.size() + vpEdgesStereo.size());
  for (size_t i = 0; i < vpEdgesMono.size(); i++) {
    MapPoint* pMP = vpEdgesMono[i].mp;
    if (pMP->isBad()) continue;
    const int32_t e = mono_edge[i];
    if (chi2[e] > 5.991 || !depth_pos[e]) vToErase.push_back(std::make_pair(vpEdgesMono[i].kf, pMP));
  }
  for (size_t i = 0; i < vpEdgesStereo.size(); i++) {
    MapPoint* pMP = vpEdgesStereo[i].mp;
    if (pMP->isBad()) continue;
    const int32_t e = stereo_edge[i];
    if (chi2[e] > 7.815 || !depth_pos[e]) vToErase.push_back(std::make_pair(vpEdgesStereo[i].kf, pMP));
  }
  // ---- write-back under the map mutex (:2001-2039)
  std::unique_lock<std::mutex> lock(pMap->mMutexMapUpdate);
  for (size_t i = 0; i < vToErase.size(); i++) {
    KeyFrame* pKFi = vToErase[i].first;
    MapPoint* pMPi = vToErase[i].second;
    pKFi->EraseMapPointMatch(pMPi);
    pMPi->EraseObservation(pKFi);
  }
  {
    int32_t k = 0;
    for (KeyFrame* pKFi : lLocalKeyFrames) {  // SE3quat.rotation().cast<float>(), translation().cast<float>()
      float q[4], t[3];
      for (int c = 0; c < 4; c++) q[c] = (float)out_q[4 * (size_t)k + c];
      for (int c = 0; c < 3; c++) t[c] = (float)out_t[3 * (size_t)k + c];
      Access::set_pose(pKFi, q, t);
      k++;
    }
  }
  {
    int32_t k = 0;
    for (MapPoint* pMP : lLocalMapPoints) {
      float X[3];
      for (int c = 0; c < 3; c++) X[c] = (float)out_p[3 * (size_t)k + c];
      Access::set_world_pos(pMP, X);
      pMP->UpdateNormalAndDepth();
      k++;
    }
  }
  pMap->IncreaseChangeIndex();
}

template <class Access, class KeyFrame, class Map>
void LocalBundleAdjuster::LocalBundleAdjustment(KeyFrame* pKF, bool* pbStopFlag, Map* pMap, int& num_fixedKF, int& num_OptKF,
                                                int& num_MPs, int& num_edges) {
  using MapPoint = typename std::remove_pointer<typename decltype(pKF->GetMapPointMatches())::value_type>::type;
  gfs_host::LocalBundleAdjustment<Access, KeyFrame, MapPoint, Map>(
      [this](const gfs_lba_problem& p, gfs_lba_solution& s, const bool* stop) { return this->solve(p, s, stop); }, pKF, pbStopFlag,
      pMap, num_fixedKF, num_OptKF, num_MPs, num_edges);
}

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

// ------------------------------------------------------------------------------------------------------------------------
// ORBmatcher::SearchByProjectionWithOF(CurrentFrame, LastFrame, mask, th, winsize, F_THRESHOLD, DIST_THRESHOLD, bMono)
// (reference src/ORBmatcher.cc:2303-2497): the bookkeeping around the two forward-backward KLT passes and their F checks —
// prior projection of the last frame's map points into the current frame (:2320-2373), the occupancy mask (cv::circle filled
// discs, :2296-2302, 2326-2332), the hand-over of failed 3-D tracks to the 2-D pass (:2434-2442), the tracked key-point lists
// that the caller passes to Frame::AddPts (:2444, 2492) — on plain arrays.  The numeric parts run on the GPU through KltTracker
// and FundamentalMatcher above.  What stays with the caller (it touches Frame / MapPoint members only): mnLastFrameSeen of the
// tracked map points, track_feature_pts_, the two AddPts calls and AssignFeaturesToGrid.
// ------------------------------------------------------------------------------------------------------------------------
struct OfFrames {
  int n_last = 0;                         // LastFrame.mvpMapPoints.size() == LastFrame.mvKeys.size()
  const gfs_keypoint* last_keys = nullptr;  // LastFrame.mvKeys (cv::KeyPoint layout)
  const uint8_t* last_has_mp = nullptr;   // mvpMapPoints[i] != nullptr
  const uint8_t* last_mp_bad = nullptr;   // mp->isBad()
  const uint8_t* last_outlier = nullptr;  // LastFrame.mvbOutlier[i]
  const float* last_mp_xw = nullptr;      // [n_last][3] mp->GetWorldPos()
  int n_cur = 0;
  const gfs_keypoint* cur_keys = nullptr;  // CurrentFrame.mvKeys
  float Tcw_q[4] = {0, 0, 0, 1}, Tcw_t[3] = {0, 0, 0};  // CurrentFrame.GetPose(): unit quaternion (x, y, z, w), translation
  float fx = 0, fy = 0, cx = 0, cy = 0, min_x = 0, max_x = 0, min_y = 0, max_y = 0;  // CurrentFrame.fx ... mnMaxY
  int img_w = 0, img_h = 0;               // CurrentFrame.image.cols / rows
};
struct OfTracked {                         // one AddPts call: tracked_kps[i] continues LastFrame key-point last_index[i]
  std::vector<gfs_keypoint> kps;
  std::vector<int32_t> last_index;
};

// cv::circle(mask, pt, radius, Scalar(255), FILLED) on CV_8UC1 (OpenCV 4.5.4 imgproc/src/drawing.cpp Circle(): the midpoint
// recurrence, filled with horizontal spans, clipped to the image; the float centre converts with cvRound like Point2f -> Point)
inline void fill_circle_u8(uint8_t* img, int rows, int cols, int stride, float fx_, float fy_, int radius) {
  const int cx = (int)std::lrint(fx_), cy = (int)std::lrint(fy_);  // saturate_cast<int>(float): round half to even
  auto hline = [&](int y, int x0, int x1) {
    if (y < 0 || y >= rows) return;
    x0 = std::max(x0, 0);
    x1 = std::min(x1, cols - 1);
    for (int x = x0; x <= x1; x++) img[(size_t)y * stride + x] = 255;
  };
  int err = 0, dx = radius, dy = 0, plus = 1, minus = (radius << 1) - 1;
  while (dx >= dy) {
    hline(cy - dy, cx - dx, cx + dx);
    hline(cy + dy, cx - dx, cx + dx);
    hline(cy - dx, cx - dy, cx + dy);
    hline(cy + dx, cx - dy, cx + dy);
    dy++;
    err += plus;
    plus += 2;
    const int m = (err <= 0) - 1;
    err -= minus & m;
    dx += m;
    minus -= m & 2;
  }
}

// returns nbgood; mask: img_h x img_w bytes (stride img_w), updated like the reference's cv::Mat& mask
inline int SearchByProjectionWithOF(KltTracker& klt, FundamentalMatcher& fmat, const KltTracker::Pyramid& last_pyr,
                                    const KltTracker::Pyramid& cur_pyr, const OfFrames& in, uint8_t* mask, float F_THRESHOLD,
                                    int DIST_THRESHOLD, OfTracked& tracked3d, OfTracked& tracked2d) {
  int nbgood = 0;
  std::vector<int32_t> v3dkpids, v2dkpids;
  std::vector<float> v3dkps, v3dpriors, v2dkps, v2dpriors;  // (x, y) pairs
  tracked3d.kps.clear();
  tracked3d.last_index.clear();
  tracked2d.kps.clear();
  tracked2d.last_index.clear();
  const int W = in.img_w, H = in.img_h;
  for (int i = 0; i < in.n_cur; i++) {  // mask of the key points already extracted in the current frame (:2326-2332)
    const float x = in.cur_keys[i].x, y = in.cur_keys[i].y;
    if (x > 0 && x < W && y > 0 && y < H) mask[(size_t)(int)y * W + (int)x] = 255;  // Mat::at<uchar>(float, float) truncates
  }
  auto push2d = [&](int cnt) {
    const float u = in.last_keys[cnt].x, v = in.last_keys[cnt].y;
    v2dkps.push_back(u);
    v2dkps.push_back(v);
    v2dpriors.push_back(u);
    v2dpriors.push_back(v);
    v2dkpids.push_back(cnt);
  };
  const float qx = in.Tcw_q[0], qy = in.Tcw_q[1], qz = in.Tcw_q[2], qw = in.Tcw_q[3];
  for (int cnt = 0; cnt < in.n_last; cnt++) {
    if (!in.last_has_mp[cnt]) {  // key points without a map point are tracked in the image only
      push2d(cnt);
      continue;
    }
    if (in.last_mp_bad[cnt] || in.last_outlier[cnt]) continue;
    // x3Dc = Tcw * x3Dw (Sophus: unit_quaternion()._transformVector in float: v + w t + q x t, t = 2 q x v)
    const float* X = in.last_mp_xw + 3 * (size_t)cnt;
    const float tx = 2.f * (qy * X[2] - qz * X[1]), ty = 2.f * (qz * X[0] - qx * X[2]), tz = 2.f * (qx * X[1] - qy * X[0]);
    const float xc = X[0] + qw * tx + (qy * tz - qz * ty) + in.Tcw_t[0];
    const float yc = X[1] + qw * ty + (qz * tx - qx * tz) + in.Tcw_t[1];
    const float zc = X[2] + qw * tz + (qx * ty - qy * tx) + in.Tcw_t[2];
    const float invzc = (float)(1.0 / (double)zc);  // `const float invzc = 1.0 / x3Dc(2)`
    const float u = in.fx * xc * invzc + in.cx, v = in.fy * yc * invzc + in.cy;
    if (invzc < 0 || u < in.min_x || u > in.max_x || v < in.min_y || v > in.max_y) {
      push2d(cnt);
      continue;
    }
    v3dkps.push_back(in.last_keys[cnt].x);
    v3dkps.push_back(in.last_keys[cnt].y);
    v3dpriors.push_back(u);
    v3dpriors.push_back(v);
    v3dkpids.push_back(cnt);
  }
  const float nklt_err = 15.f, max_fbklt_dist = 0.5f;
  auto track_and_check = [&](int nbpyrlvl, const std::vector<float>& kps, std::vector<float>& priors, float f_thr,
                             std::vector<uint8_t>& vkpstatus) {
    klt.fbKltTracking(last_pyr, cur_pyr, nbpyrlvl, nklt_err, max_fbklt_dist, kps, priors, vkpstatus);
    std::vector<int32_t> index;
    std::vector<float> un_cur, un_forw;
    for (size_t i = 0; i < vkpstatus.size(); i++)
      if (vkpstatus[i]) {
        index.push_back((int32_t)i);
        un_cur.push_back(kps[2 * i]);
        un_cur.push_back(kps[2 * i + 1]);
        un_forw.push_back(priors[2 * i]);
        un_forw.push_back(priors[2 * i + 1]);
      }
    if (index.size() > 8) {  // cv::findFundamentalMat(un_cur_pts, un_forw_pts, FM_RANSAC, f_thr, 0.99, status)
      std::vector<uint8_t> status;
      fmat.findFundamentalMat(un_cur, un_forw, (double)f_thr, 0.99, status);
      for (size_t i = 0; i < status.size(); i++)
        if (!status[i]) vkpstatus[(size_t)index[i]] = 0;
    }
  };
  // cv::Point(pt.x, pt.y): truncation.  A track that left the image cannot sit next to a key point of the mask: it is not "nearby"
  // (the reference reads mask.at<uchar>() unchecked behind cv's own in-image filtering; here nothing is read outside the buffer).
  auto is_nearby = [&](float x, float y) {
    const int xi = (int)x, yi = (int)y;
    return xi >= 0 && xi < W && yi >= 0 && yi < H && mask[(size_t)yi * W + xi] == 255;
  };
  if (!v3dkpids.empty()) {  // 1st: key points with a map point, prior = projection (3 pyramid levels)
    std::vector<uint8_t> vkpstatus;
    track_and_check(3, v3dkps, v3dpriors, F_THRESHOLD, vkpstatus);
    for (size_t i = 0; i < v3dkpids.size(); i++) {
      if (vkpstatus[i]) {
        gfs_keypoint pt = in.last_keys[v3dkpids[i]];
        pt.x = v3dpriors[2 * i];
        pt.y = v3dpriors[2 * i + 1];
        if (is_nearby(pt.x, pt.y)) continue;
        tracked3d.kps.push_back(pt);
        tracked3d.last_index.push_back(v3dkpids[i]);
        nbgood++;
        fill_circle_u8(mask, H, W, W, pt.x, pt.y, DIST_THRESHOLD);
      } else {
        push2d(v3dkpids[i]);  // not tracked: tried again as a 2-D point
      }
    }
  }
  if (!v2dkpids.empty()) {  // 2nd: image-only tracking (6 pyramid levels requested), F threshold halved
    std::vector<uint8_t> vkpstatus;
    track_and_check(6, v2dkps, v2dpriors, F_THRESHOLD * 0.5f, vkpstatus);
    for (size_t i = 0; i < v2dkpids.size(); i++)
      if (vkpstatus[i]) {
        gfs_keypoint pt = in.last_keys[v2dkpids[i]];
        pt.x = v2dpriors[2 * i];
        pt.y = v2dpriors[2 * i + 1];
        if (is_nearby(pt.x, pt.y)) continue;
        tracked2d.kps.push_back(pt);
        tracked2d.last_index.push_back(v2dkpids[i]);
        fill_circle_u8(mask, H, W, W, pt.x, pt.y, DIST_THRESHOLD);
        nbgood++;
      }
  }
  return nbgood;
}

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

// The drop-ins written against the reference's own types (cv::Mat, cv::KeyPoint, Eigen, Sophus, ORB_SLAM3::ORBextractor as a base
// class, small_gicp::RegistrationResult) are in gfs_reference_dropins.hpp beside this file.

