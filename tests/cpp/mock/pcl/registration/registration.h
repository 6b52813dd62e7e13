// TEST DOUBLE of pcl::Registration -- see tests/cpp/mock/README.md.  Members and virtuals as
// the adapter's PCL face uses them (PCL 1.14: registration.h).
#pragma once
#include <Eigen/Core>
#include <chrono>
#include <string>
#include <vector>
#include "../point_cloud.h"
namespace pcl {
template <typename PointSource, typename PointTarget, typename Scalar = float>
class Registration {
 public:
  using Matrix4 = Eigen::Matrix<Scalar, 4, 4>;
  using Ptr = shared_ptr<Registration<PointSource, PointTarget, Scalar>>;
  using ConstPtr = shared_ptr<const Registration<PointSource, PointTarget, Scalar>>;
  using PointCloudSource = pcl::PointCloud<PointSource>;
  using PointCloudSourcePtr = typename PointCloudSource::Ptr;
  using PointCloudSourceConstPtr = typename PointCloudSource::ConstPtr;
  using PointCloudTarget = pcl::PointCloud<PointTarget>;
  using PointCloudTargetPtr = typename PointCloudTarget::Ptr;
  using PointCloudTargetConstPtr = typename PointCloudTarget::ConstPtr;

  Registration() { final_transformation_ = Matrix4::Identity(); transformation_ = Matrix4::Identity(); }
  virtual ~Registration() = default;
  virtual void setInputSource(const PointCloudSourceConstPtr& cloud) { input_ = cloud; source_cloud_updated_ = true; }
  virtual void setInputTarget(const PointCloudTargetConstPtr& cloud) { target_ = cloud; target_cloud_updated_ = true; }
  Matrix4 getFinalTransformation() { return final_transformation_; }
  bool hasConverged() const { return converged_; }
  void setMaximumIterations(int n) { max_iterations_ = n; }
  void setTransformationEpsilon(double e) { transformation_epsilon_ = e; }
  void align(PointCloudSource& output) { align(output, Matrix4::Identity()); }
  // What the real pcl::Registration::align does AROUND computeTransformation (PCL 1.14 registration.hpp:
  // align(), and PCLBase::initCompute) is reproduced step for step, because a drop-in engine pays it on every
  // align() through RegisterCallback::registration (align is not virtual; setFillOutputCloud(false) on the
  // adapter cannot remove it): the identity index vector, output.resize + header fields, a per-point copy of
  // the source through the indices, and a second pass that sets data[3] = 1.
  void align(PointCloudSource& output, const Matrix4& guess) {
    if (!target_ || !input_) return;          // initCompute()
    kdtree_builds_ += target_cloud_updated_;  // the real one builds a FLANN tree over the target here
    target_cloud_updated_ = false;
    const auto t0 = std::chrono::steady_clock::now();
    if (indices_.size() != input_->size()) {  // PCLBase::initCompute: fake indices 0 .. n-1
      indices_.resize(input_->size());
      for (size_t i = 0; i < indices_.size(); ++i) indices_[i] = static_cast<int>(i);
    }
    output.resize(indices_.size());
    output.width = input_->width;
    output.height = input_->height;
    output.is_dense = input_->is_dense;
    for (size_t i = 0; i < indices_.size(); ++i) output[i] = (*input_)[indices_[i]];
    converged_ = false;
    final_transformation_ = transformation_ = Matrix4::Identity();
    for (size_t i = 0; i < indices_.size(); ++i) output[i].pad = 1.0f;  // data[3] = 1
    prework_ns_ += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
    ++prework_calls_;
    computeTransformation(output, guess);
  }
  long long prework_ns_ = 0, prework_calls_ = 0;  // mock-only probe: time spent before computeTransformation
  int kdtree_builds_ = 0;  // mock-only probe

 protected:
  virtual void computeTransformation(PointCloudSource& output, const Matrix4& guess) = 0;
  std::string reg_name_;
  int nr_iterations_ = 0, max_iterations_ = 10;
  PointCloudSourceConstPtr input_;
  PointCloudTargetConstPtr target_;
  Matrix4 final_transformation_, transformation_;
  double transformation_epsilon_ = 0.0;
  bool converged_ = false;
  bool target_cloud_updated_ = true, source_cloud_updated_ = true;
  std::vector<int> indices_;  // pcl::PCLBase::indices_
};
}  // namespace pcl
