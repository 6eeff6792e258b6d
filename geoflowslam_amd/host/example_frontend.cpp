// Minimal C++ host program over the C ABI (no OpenCV / Eigen): extract ORB on two synthetic frames, match,
// register two small clouds (plain and streaming form), then the optical-flow stream: track the key points of frame 0 into
// frame 1 (forward-backward checked) and run the fundamental-matrix check on the tracks.  Build on an MI355X box:
//   g++ -std=c++17 example_frontend.cpp -I../../include -L.. -lgfs_hip -Wl,-rpath,'$ORIGIN/..' -o example_frontend
#include <cmath>
#include <cstdio>

#include "gfs_adaptors.hpp"

int main() {
  const int W = 640, H = 480;
  std::vector<uint8_t> img0((size_t)W * H), img1((size_t)W * H);
  for (int y = 0; y < H; y++)
    for (int x = 0; x < W; x++)
      img0[(size_t)y * W + x] = (uint8_t)(128 + 60 * ((x / 23 + y / 17) % 2) + 30 * ((x / 7 + y / 5) % 2) - 45);
  for (int y = 0; y < H; y++)  // the same scene seen 3 pixels to the left
    for (int x = 0; x < W; x++) img1[(size_t)y * W + x] = img0[(size_t)y * W + (x + 3) % W];
  try {
    gfs_host::ORBextractor ext(1000, 1.2f, 8, 20, 7, H, W);
    std::vector<gfs_keypoint> k0, k1;
    std::vector<uint8_t> d0, d1;
    std::vector<int> lap = {0, 0};
    const int m0 = ext(img0.data(), H, W, W, k0, d0, lap), m1 = ext(img1.data(), H, W, W, k1, d1, lap);
    gfs_host::ORBmatcher matcher;
    std::vector<gfs_host::DMatch> matches;
    matcher.match(d0.data(), (int)k0.size(), d1.data(), (int)k1.size(), matches);
    std::printf("ORB: %d / %d keypoints (mono %d / %d), %zu matches\n", (int)k0.size(), (int)k1.size(), m0, m1, matches.size());
    std::vector<float> c0, c1;
    for (int i = 0; i < 60; i++)
      for (int j = 0; j < 60; j++) {
        const float x = 0.03f * i, y = 0.03f * j, z = 2.f + 0.1f * std::sin(3 * x) * std::cos(2 * y);
        c0.insert(c0.end(), {x, y, z, 1.f});
        c1.insert(c1.end(), {x - 0.01f, y + 0.005f, z, 1.f});
      }
    const double I[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    gfs_host::RegistrationGICP reg;
    const gfs_gicp_result r = reg.RegisterPointClouds(c0.data(), 3600, c1.data(), 3600, I);
    std::printf("GICP: converged=%d inliers=%llu t=(%.4f %.4f %.4f)\n", r.converged, (unsigned long long)r.num_inliers,
                r.T_target_source[12], r.T_target_source[13], r.T_target_source[14]);
    const gfs_gicp_result r2 = reg.RegisterNext(c0.data(), 3600, I);  // target = the cloud registered last (c1), kept on the device
    std::printf("GICP (streaming): converged=%d inliers=%llu t=(%.4f %.4f %.4f)\n", r2.converged, (unsigned long long)r2.num_inliers,
                r2.T_target_source[12], r2.T_target_source[13], r2.T_target_source[14]);
    // optical flow: Frame::mImGray pyramids, ORBmatcher::fbKltTracking, cv::findFundamentalMat(FM_RANSAC)
    gfs_host::KltTracker klt(W, H, 21);
    gfs_host::KltTracker::Pyramid p0(klt), p1(klt);
    p0.build(img0.data(), W);
    p1.build(img1.data(), W);
    std::vector<float> kps, priors;
    for (const gfs_keypoint& k : k0) {
      kps.insert(kps.end(), {k.x, k.y});
      priors.insert(priors.end(), {k.x, k.y});
    }
    std::vector<uint8_t> status;
    klt.fbKltTracking(p0, p1, 3, 15.f, 0.5f, kps, priors, status);
    std::vector<float> a, b;
    for (size_t i = 0; i < status.size(); i++)
      if (status[i]) {
        a.insert(a.end(), {kps[2 * i], kps[2 * i + 1]});
        b.insert(b.end(), {priors[2 * i], priors[2 * i + 1]});
      }
    std::printf("KLT: %zu of %zu key points tracked forward and back", a.size() / 2, status.size());
    if (a.size() / 2 > 8) {
      gfs_host::FundamentalMatcher fm;
      std::vector<uint8_t> inl;
      const int n_in = fm.findFundamentalMat(a, b, 1.0, 0.99, inl);
      std::printf("; F check keeps %d", n_in);
    }
    std::printf("\n");
  } catch (const std::exception& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}
