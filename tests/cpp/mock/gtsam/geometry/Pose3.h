// TEST DOUBLE of gtsam::Pose3 -- see tests/cpp/mock/README.md
#pragma once
#include <Eigen/Core>
namespace gtsam {
class Pose3 {
 public:
  Pose3() : T_(Eigen::Matrix4d::Identity()) {}
  explicit Pose3(const Eigen::Matrix4d& T) : T_(T) {}
  Eigen::Matrix4d matrix() const { return T_; }
  Eigen::Vector3d translation() const { Eigen::Vector3d t; for (int i = 0; i < 3; ++i) t(i) = T_(i, 3); return t; }
 private:
  Eigen::Matrix4d T_;
};
}  // namespace gtsam
