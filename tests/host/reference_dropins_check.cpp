// Syntax / layout check of geoflowslam_amd/host/gfs_reference_dropins.hpp (g++ -fsyntax-only; never linked): the reference's real
// ORBextractor.h and registration_result.hpp (-I /root/reference/...) over declaration-only OpenCV / Eigen / Sophus stand-ins.
#include <opencv2/opencv.hpp>
#include <Eigen/Geometry>
#include <sophus/se3.hpp>

#ifdef GFS_HAVE_REFERENCE_TREE
#include <ORBextractor.h>
#include <small_gicp/registration/registration_result.hpp>
#else  // the reference tree is not on this machine: the same two declarations, reduced to what the drop-ins use
namespace ORB_SLAM3 {
class ORBextractor {
 public:
  ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST);
  virtual ~ORBextractor() {}
  virtual int operator()(cv::InputArray _image, cv::InputArray _mask, std::vector<cv::KeyPoint>& _keypoints, cv::OutputArray _descriptors,
                         std::vector<int>& vLappingArea);
  int inline GetLevels() { return nlevels; }
  std::vector<cv::Mat> mvImagePyramid;

 protected:
  int nlevels;
};
}  // namespace ORB_SLAM3
namespace small_gicp {
struct RegistrationResult {
  RegistrationResult(const Eigen::Isometry3d& T = Eigen::Isometry3d::Identity());
  Eigen::Isometry3d T_target_source;
  bool converged;
  size_t iterations, num_inliers;
  Eigen::Matrix<double, 6, 6> H;
  Eigen::Matrix<double, 6, 1> b;
  double error;
};
}  // namespace small_gicp
#endif

#include "../../geoflowslam_amd/host/gfs_reference_dropins.hpp"

// the members of the reference's KeyFrame / MapPoint the Access policy touches (include/KeyFrame.h, include/MapPoint.h)
struct KeyFrameLike {
  Sophus::SE3f GetPose();
  void SetPose(const Sophus::SE3f&);
};
struct MapPointLike {
  Eigen::Vector3f GetWorldPos();
  void SetWorldPos(const Eigen::Vector3f&);
};

void instantiate_everything() {
  std::vector<int> lap{0, 0};
  gfs_dropin::GfsORBextractor ext(1000, 1.2f, 8, 20, 7);
  ORB_SLAM3::ORBextractor* base = &ext;  // the factory returns the base pointer (src/ORBextractor.cc:1253-1265)
  cv::Mat im, desc, d2;
  std::vector<cv::KeyPoint> kps, kps2;
  ext.fill_image_pyramid = true;  // the stereo matcher reads mvImagePyramid (src/Frame.cc:1159-1256)
  (*base)(im, cv::Mat(), kps, desc, lap);  // Frame::ExtractORB, src/Frame.cc:768-777
  std::vector<cv::DMatch> matches;
  gfs_dropin::bf_match(desc, d2, matches);
  std::vector<bool> inl;
  (void)gfs_dropin::gms_inlier_mask(kps, kps2, cv::Size(640, 480), matches, inl);
  std::vector<Eigen::Vector4f> a, b;
  small_gicp::RegistrationResult r = gfs_dropin::RegisterPointClouds(a, b, Eigen::Isometry3d::Identity());
  (void)r;
  using A = gfs_dropin::LbaAccess<KeyFrameLike, MapPointLike>;
  KeyFrameLike kf;
  MapPointLike mp;
  float q[4], t[3], x[3];
  A::pose(&kf, q, t);
  A::set_pose(&kf, q, t);
  A::world_pos(&mp, x);
  A::set_world_pos(&mp, x);
}
