// Test harness for gfs_host::SearchByProjectionWithOF (geoflowslam_amd/host/gfs_adaptors.hpp): runs the adaptor (GPU KLT + F
// check underneath) on flat inputs and returns the two tracked lists, nbgood and the updated mask.  Built by
// tests/test_gpu_sbp_of.py with g++.
#include <cstring>

#include "../../geoflowslam_amd/host/gfs_adaptors.hpp"

extern "C" int sbp_of_test(const uint8_t* last_img, const uint8_t* cur_img, int w, int h, int winsize, int n_last,
                           const gfs_keypoint* last_keys, const uint8_t* has_mp, const uint8_t* mp_bad, const uint8_t* outlier,
                           const float* xw, int n_cur, const gfs_keypoint* cur_keys, const float* Tcw_q, const float* Tcw_t,
                           const float* K8 /*fx fy cx cy minx maxx miny maxy*/, float F_THRESHOLD, int DIST_THRESHOLD, uint8_t* mask,
                           gfs_keypoint* out3d, int32_t* idx3d, int32_t* n3d, gfs_keypoint* out2d, int32_t* idx2d, int32_t* n2d) {
  try {
    gfs_host::KltTracker klt(w, h, winsize, 3, 8192);
    gfs_host::FundamentalMatcher fm(8192);
    gfs_host::KltTracker::Pyramid p0(klt), p1(klt);
    p0.build(last_img, w);
    p1.build(cur_img, w);
    gfs_host::OfFrames in;
    in.n_last = n_last;
    in.last_keys = last_keys;
    in.last_has_mp = has_mp;
    in.last_mp_bad = mp_bad;
    in.last_outlier = outlier;
    in.last_mp_xw = xw;
    in.n_cur = n_cur;
    in.cur_keys = cur_keys;
    std::memcpy(in.Tcw_q, Tcw_q, 16);
    std::memcpy(in.Tcw_t, Tcw_t, 12);
    in.fx = K8[0];
    in.fy = K8[1];
    in.cx = K8[2];
    in.cy = K8[3];
    in.min_x = K8[4];
    in.max_x = K8[5];
    in.min_y = K8[6];
    in.max_y = K8[7];
    in.img_w = w;
    in.img_h = h;
    gfs_host::OfTracked t3, t2;
    const int good = gfs_host::SearchByProjectionWithOF(klt, fm, p0, p1, in, mask, F_THRESHOLD, DIST_THRESHOLD, t3, t2);
    *n3d = (int32_t)t3.kps.size();
    *n2d = (int32_t)t2.kps.size();
    if (*n3d) {
      std::memcpy(out3d, t3.kps.data(), t3.kps.size() * sizeof(gfs_keypoint));
      std::memcpy(idx3d, t3.last_index.data(), t3.last_index.size() * 4);
    }
    if (*n2d) {
      std::memcpy(out2d, t2.kps.data(), t2.kps.size() * sizeof(gfs_keypoint));
      std::memcpy(idx2d, t2.last_index.data(), t2.last_index.size() * 4);
    }
    return good;
  } catch (const std::exception& ex) {
    fprintf(stderr, "sbp_of_test: %s\n", ex.what());
    return -1;
  }
}

extern "C" void fill_circle_test(uint8_t* img, int rows, int cols, float x, float y, int radius) {
  gfs_host::fill_circle_u8(img, rows, cols, cols, x, y, radius);
}
