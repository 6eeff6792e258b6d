// Test harness for the plain-container mirror of the reference's classes (namespace gfs_host, geoflowslam_amd/host/gfs_adaptors.hpp):
// ORBextractor::operator(), ORBmatcher::DescriptorDistance / match, RegistrationGICP::RegisterPointClouds / RegisterNext and
// GmsMatcher::GetInlierMask, ProjectionMatcher::SearchByProjection and PoseOptimizer::PoseOptimization are driven from C++ exactly as a maintainer's adaptor would drive them, and hand their results back to
// the Python test (tests/test_gpu_host_mirror.py), which compares them with the oracle.
#include <cstring>

#include "../../geoflowslam_amd/host/gfs_adaptors.hpp"

extern "C" {

// returns monoIndex (or -1), *n = key-points written
int hm_orb(const uint8_t* image, int rows, int cols, int stride, int nfeatures, float scale, int nlevels, int ini_th, int min_th,
           int lap0, int lap1, gfs_keypoint* kps, uint8_t* desc, int cap, int* n, float* tables /*[4][nlevels]*/) {
  try {
    gfs_host::ORBextractor ext(nfeatures, scale, nlevels, ini_th, min_th, rows > 0 ? rows : 480, cols > 0 ? cols : 640);
    std::vector<gfs_keypoint> k;
    std::vector<uint8_t> d;
    std::vector<int> lap;
    if (lap0 || lap1) lap = {lap0, lap1};
    const int mono = ext(image, rows, cols, stride, k, d, lap);
    *n = (int)k.size();
    if ((int)k.size() > cap) return -1000;
    if (!k.empty()) memcpy(kps, k.data(), k.size() * sizeof(gfs_keypoint));
    if (!d.empty()) memcpy(desc, d.data(), d.size());
    if (tables) {
      const std::vector<float> t[4] = {ext.GetScaleFactors(), ext.GetInverseScaleFactors(), ext.GetScaleSigmaSquares(),
                                       ext.GetInverseScaleSigmaSquares()};
      if (ext.GetLevels() != nlevels) return -1001;
      for (int w = 0; w < 4; w++) memcpy(tables + (size_t)w * nlevels, t[w].data(), (size_t)nlevels * sizeof(float));
    }
    return mono;
  } catch (const std::exception&) {
    return -2000;
  }
}

// matches as (queryIdx, trainIdx, distance) triples; returns their number
int hm_match(const uint8_t* q, int nq, const uint8_t* t, int nt, int* query_idx, int* train_idx, float* dist, int* dd_first) {
  try {
    gfs_host::ORBmatcher m;
    std::vector<gfs_host::DMatch> out;
    m.match(q, nq, t, nt, out);
    for (size_t i = 0; i < out.size(); i++) {
      query_idx[i] = out[i].queryIdx;
      train_idx[i] = out[i].trainIdx;
      dist[i] = out[i].distance;
    }
    if (dd_first && nq > 0 && nt > 0) *dd_first = gfs_host::ORBmatcher::DescriptorDistance(q, t);
    return (int)out.size();
  } catch (const std::exception&) {
    return -2000;
  }
}

// two consecutive registrations: (a -> b) with RegisterPointClouds, then (b -> c) with RegisterNext
int hm_gicp(const float* a, int na, const float* b, int nb, const float* c, int nc, const double* init16, gfs_gicp_result* r_ab,
            gfs_gicp_result* r_bc) {
  try {
    gfs_host::RegistrationGICP reg(65536);
    *r_ab = reg.RegisterPointClouds(a, na, b, nb, init16);
    if (c) *r_bc = reg.RegisterNext(c, nc, init16);
    return 0;
  } catch (const std::exception&) {
    return -2000;
  }
}

int hm_gms(const gfs_keypoint* k1, int n1, int w1, int h1, const gfs_keypoint* k2, int n2, int w2, int h2, const int* qi, const int* ti,
           int nm, uint8_t* mask) {
  try {
    gfs_host::GmsMatcher g;
    std::vector<int32_t> q(qi, qi + nm), t(ti, ti + nm);
    std::vector<uint8_t> m;
    const int nin = g.GetInlierMask(k1, n1, w1, h1, k2, n2, w2, h2, q, t, m);
    if (nm) memcpy(mask, m.data(), (size_t)nm);
    return nin;
  } catch (const std::exception&) {
    return -2000;
  }
}

int hm_sbp(const gfs_sbp_problem* p, int32_t* cur_match) {
  try {
    gfs_host::ProjectionMatcher pm(8192, 4096);
    std::vector<int32_t> m;
    const int n = pm.SearchByProjection(*p, m);
    if (!m.empty()) memcpy(cur_match, m.data(), m.size() * sizeof(int32_t));
    return n;
  } catch (const std::exception&) {
    return -2000;
  }
}

int hm_pose(const gfs_pose_problem* p, gfs_pose_solution* s) {
  try {
    gfs_host::PoseOptimizer po(8192);
    return po.PoseOptimization(*p, *s);
  } catch (const std::exception&) {
    return -2000;
  }
}

}  // extern "C"
