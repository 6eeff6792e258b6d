// TEST INFRASTRUCTURE — NOT PRODUCT CODE (see gfs_oracle.h header). PARITY UNPINNED.
// CPU restatement of Frame::ConvertDepthToPointCloud (reference src/Frame.cc:590-623) and
// Frame::ComputeStereoFromRGBD (src/Frame.cc:1314-1332).
#include <cstddef>
#include <cstdint>
using std::size_t;

#include "gfs_oracle.h"

extern "C" {

int gfso_depth_to_cloud(const float* depth, int rows, int cols, int stride_elems, int downsample, float fx, float fy, float cx,
                        float cy, float* out_xyzw, int cap) {
  if (!depth || rows <= 0 || cols <= 0) return 0;  // "Error: Depth image is empty."
  int n = 0;
  for (int v = 0; v < rows; v += downsample) {
    for (int u = 0; u < cols; u += downsample) {
      float d = depth[(size_t)v * stride_elems + u];
      if (d > 0.0 && d < 10.0) {
        float x = (u - cx) * d / fx;
        float y = (v - cy) * d / fy;
        if (n < cap) {
          out_xyzw[4 * (size_t)n] = x;
          out_xyzw[4 * (size_t)n + 1] = y;
          out_xyzw[4 * (size_t)n + 2] = d;
          out_xyzw[4 * (size_t)n + 3] = 1.0f;
        }
        n++;
      }
    }
  }
  return n;
}

void gfso_stereo_from_rgbd(const gfso_keypoint* kps, const float* kps_un_x, int n, const float* depth, int stride_elems,
                           float bf, float* u_right, float* depth_out) {
  for (int i = 0; i < n; i++) {
    u_right[i] = -1;
    depth_out[i] = -1;
    const float v = kps[i].y, u = kps[i].x;
    const float d = depth[(size_t)(int)v * stride_elems + (int)u];  // imDepth.at<float>(v, u): float -> int truncation
    if (d > 0) {
      depth_out[i] = d;
      u_right[i] = (kps_un_x ? kps_un_x[i] : kps[i].x) - bf / d;
    }
  }
}

}  // extern "C"
