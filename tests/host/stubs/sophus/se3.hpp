// DECLARATION-ONLY stand-in for Sophus::SE3f (see tests/host/stubs/opencv2/core/core.hpp).
#pragma once
#include <Eigen/Geometry>
namespace Sophus {
template <typename T>
class SE3 {
 public:
  SE3();
  SE3(const Eigen::Quaternion<T>&, const Eigen::Matrix<T, 3, 1>&);
  Eigen::Quaternion<T> unit_quaternion() const;
  const Eigen::Matrix<T, 3, 1>& translation() const;
};
typedef SE3<float> SE3f;
}  // namespace Sophus
