// ndt_hip.hpp -- header-only C++ adapter over the C-ABI (include/ndt_hip.h) with the method
// names of `pclomp::NormalDistributionsTransform` that the reference's drivers call
// (ref: run/pipeline.cpp:464-481,557-568; run/pipeline_ligo_tc.cpp:287-305,529-538;
//  include/pipeline.hpp:175-206; extern/svn_ndt/test/test_svn_ndt.cpp:144-179).
//
// Two faces, same names:
//   * with PCL (`__has_include(<pcl/registration/registration.h>)`, or -DNDT_HIP_WITH_PCL=1):
//     `ndt_hip::NormalDistributionsTransform<PointSource, PointTarget>` derives from
//     `pcl::Registration`, so it can be assigned to `RegisterCallback::registration`
//     (ref: include/registercallback.hpp:35) and driven through setInputTarget /
//     setInputSource / align / getFinalTransformation / hasConverged unchanged.
//     NOTE: this branch cannot be compiled in the build image (no PCL/Eigen there).
//   * without PCL: a dependency-free twin on `ndt_hip::PointCloud<PointT>` and
//     `ndt_hip::Matrix4f` (16 floats, column-major) used by tests/cpp/test_adapter.cpp.
// Failures never throw: like the reference (ref: svn_ndt_impl.hpp:682-702) a failed align
// leaves the guess as the final transformation and hasConverged() == false; the status and
// message are available through lastStatus() / lastError().
#pragma once

#include <array>
#include <cstddef>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../ndt_hip.h"

#if !defined(NDT_HIP_WITH_PCL)
#if defined(__has_include)
#if __has_include(<pcl/registration/registration.h>)
#define NDT_HIP_WITH_PCL 1
#endif
#endif
#endif
#ifndef NDT_HIP_WITH_PCL
#define NDT_HIP_WITH_PCL 0
#endif

#if NDT_HIP_WITH_PCL
#include <pcl/point_cloud.h>
#include <pcl/registration/registration.h>
#include <Eigen/Core>
#endif

namespace ndt_hip {

// same order as pclomp::NeighborSearchMethod
enum NeighborSearchMethod { KDTREE = NDT_KDTREE, DIRECT26 = NDT_DIRECT26, DIRECT7 = NDT_DIRECT7, DIRECT1 = NDT_DIRECT1 };

using Matrix4f = std::array<float, 16>;  // column-major, Eigen::Matrix4f layout

inline Matrix4f identity4f() {
  Matrix4f m{};
  m[0] = m[5] = m[10] = m[15] = 1.0f;
  return m;
}

// mirrors pclomp::NdtResult (fields the drivers read: iteration_num, hessian)
struct NdtResult {
  Matrix4f pose = identity4f();
  float transform_probability = 0.0f;
  float nearest_voxel_transformation_likelihood = 0.0f;
  int iteration_num = 0;
  std::array<double, 36> hessian{};  // row-major 6x6 of the maximised score
  // -(hessian + eps I)^-1 in the block order the drivers hand to GTSAM
  // (ref: run/pipeline.cpp:594-603, src/registercallback.cpp:170-186); false if singular
  bool covarianceForGtsam(std::array<double, 36>& cov, double eps = 1e-6, bool gtsam_order = true) const {
    return ndt_result_covariance(hessian.data(), eps, gtsam_order ? 1 : 0, cov.data()) == NDT_OK;
  }
#if NDT_HIP_WITH_PCL
  Eigen::Matrix4f poseEigen() const { return Eigen::Map<const Eigen::Matrix4f>(pose.data()); }
  Eigen::Matrix<double, 6, 6> hessianEigen() const {
    return Eigen::Map<const Eigen::Matrix<double, 6, 6, Eigen::RowMajor>>(hessian.data());
  }
#endif
};

// getTargetCells(): the accessors extractNdtData() uses (ref: include/pipeline.hpp:175-206)
class TargetGrid {
 public:
  struct Leaf {
    ndt_leaf d;
    int getPointCount() const { return d.point_count; }
    const double* getMean() const { return d.mean; }
    const double* getCov() const { return d.cov; }
    const double* getInverseCov() const { return d.icov; }
    const double* getEvecs() const { return d.evecs; }
    const double* getEvals() const { return d.evals; }
  };
  struct Entry {
    size_t first;  // voxel index
    Leaf second;
  };
  const std::vector<Entry>& getLeaves() const { return leaves_; }
  int getMinPointPerVoxel() const { return min_points_; }
  std::array<float, 3> getLeafCenter(size_t index) const {
    for (const Entry& e : leaves_)
      if (e.first == index) return {e.second.d.center[0], e.second.d.center[1], e.second.d.center[2]};
    return {0.0f, 0.0f, 0.0f};
  }
  std::vector<Entry> leaves_;
  int min_points_ = 6;
};

#if NDT_HIP_WITH_PCL
template <typename PointT>
using PointCloud = pcl::PointCloud<PointT>;
#else
template <typename PointT>
struct PointCloud {
  using Ptr = std::shared_ptr<PointCloud<PointT>>;
  using ConstPtr = std::shared_ptr<const PointCloud<PointT>>;
  std::vector<PointT> points;
  size_t size() const { return points.size(); }
  bool empty() const { return points.empty(); }
};
struct PointXYZ { float x, y, z, pad; };                        // 16 bytes like pcl::PointXYZ
struct PointXYZI { float x, y, z, pad; float intensity, p1, p2, p3; };  // 32 bytes like pcl::PointXYZI
#endif

template <typename PointSource, typename PointTarget>
class NormalDistributionsTransform
#if NDT_HIP_WITH_PCL
    : public pcl::Registration<PointSource, PointTarget>
#endif
{
 public:
  using PointCloudSource = PointCloud<PointSource>;
  using PointCloudTarget = PointCloud<PointTarget>;
  using Ptr = std::shared_ptr<NormalDistributionsTransform<PointSource, PointTarget>>;
  using ConstPtr = std::shared_ptr<const NormalDistributionsTransform<PointSource, PointTarget>>;

  NormalDistributionsTransform() {
    ndt_default_params(&prm_);
    status_ = ndt_create(&prm_, &h_);
#if NDT_HIP_WITH_PCL
    this->reg_name_ = "ndt_hip::NormalDistributionsTransform";
    this->max_iterations_ = prm_.max_iterations;
    this->transformation_epsilon_ = prm_.trans_epsilon;
#endif
  }
  ~NormalDistributionsTransform() { ndt_destroy(h_); }
  NormalDistributionsTransform(const NormalDistributionsTransform&) = delete;
  NormalDistributionsTransform& operator=(const NormalDistributionsTransform&) = delete;

  // ---- pclomp setters (ref: run/pipeline.cpp:467-480, run/pipeline_ligo_tc.cpp:293) ----
  void setNumThreads(int n) { prm_.num_threads = n; push(); }
  int getNumThreads() const { return prm_.num_threads; }
  void setResolution(float r) { prm_.resolution = r; push(); }
  float getResolution() const { return prm_.resolution; }
  void setStepSize(double s) { prm_.step_size = s; push(); }
  double getStepSize() const { return prm_.step_size; }
  void setOutlierRatio(double o) { prm_.outlier_ratio = o; push(); }
  double getOulierRatio() const { return prm_.outlier_ratio; }  // sic, the PCL spelling
  void setNeighborhoodSearchMethod(NeighborSearchMethod m) { prm_.search_method = (int)m; push(); }
  void setRegularizationScaleFactor(float k) { prm_.regularization_scale_factor = k; push(); }
  void setMinPointPerVoxel(int n) { prm_.min_points_per_voxel = n; push(); }
#if NDT_HIP_WITH_PCL
  void setTransformationEpsilon(double e) { this->transformation_epsilon_ = e; prm_.trans_epsilon = e; push(); }
  void setMaximumIterations(int n) { this->max_iterations_ = n; prm_.max_iterations = n; push(); }
  void setRegularizationPose(const Eigen::Matrix4f& T) { status_ = ndt_set_regularization_pose(h_, T.data()); }
#else
  void setTransformationEpsilon(double e) { prm_.trans_epsilon = e; push(); }
  void setMaximumIterations(int n) { prm_.max_iterations = n; push(); }
  void setRegularizationPose(const Matrix4f& T) { status_ = ndt_set_regularization_pose(h_, T.data()); }
#endif
  void unsetRegularizationPose() { status_ = ndt_clear_regularization_pose(h_); }

  // ---- clouds ----
#if NDT_HIP_WITH_PCL
  void setInputTarget(const typename pcl::Registration<PointSource, PointTarget>::PointCloudTargetConstPtr& cloud) override {
    pcl::Registration<PointSource, PointTarget>::setInputTarget(cloud);
    uploadTarget(cloud.get());
  }
  void setInputSource(const typename pcl::Registration<PointSource, PointTarget>::PointCloudSourceConstPtr& cloud) override {
    pcl::Registration<PointSource, PointTarget>::setInputSource(cloud);
    uploadSource(cloud.get());
  }
#else
  void setInputTarget(const typename PointCloudTarget::ConstPtr& cloud) { uploadTarget(cloud.get()); }
  void setInputSource(const typename PointCloudSource::ConstPtr& cloud) { source_ = cloud; uploadSource(cloud.get()); }
#endif

  // ---- registration ----
#if NDT_HIP_WITH_PCL
  // called by pcl::Registration::align(output, guess)
  void computeTransformation(PointCloudSource& output, const Eigen::Matrix4f& guess) override {
    run(guess.data());
    this->final_transformation_ = Eigen::Map<const Eigen::Matrix4f>(res_.final_transformation);
    this->transformation_ = this->final_transformation_;
    this->converged_ = res_.converged != 0;
    this->nr_iterations_ = res_.iterations;
    fillOutput(output);
  }
#else
  void computeTransformation(PointCloudSource& output, const Matrix4f& guess) {
    run(guess.data());
    fillOutput(output);
  }
  void align(PointCloudSource& output, const Matrix4f& guess = identity4f()) { computeTransformation(output, guess); }
  Matrix4f getFinalTransformation() const {
    Matrix4f m;
    std::memcpy(m.data(), res_.final_transformation, sizeof(float) * 16);
    return m;
  }
  bool hasConverged() const { return res_.converged != 0; }
#endif
  int getFinalNumIteration() const { return res_.iterations; }
  double getTransformationProbability() const { return res_.transform_probability; }
  double getNearestVoxelTransformationLikelihood() const { return res_.nearest_voxel_transformation_likelihood; }

  NdtResult getResult() const {
    NdtResult r;
    std::memcpy(r.pose.data(), res_.final_transformation, sizeof(float) * 16);
    r.transform_probability = (float)res_.transform_probability;
    r.nearest_voxel_transformation_likelihood = (float)res_.nearest_voxel_transformation_likelihood;
    r.iteration_num = res_.iterations;
    std::memcpy(r.hessian.data(), res_.hessian, sizeof(double) * 36);
    return r;
  }

  // ---- voxel grid (ref: include/pipeline.hpp:178-206) ----
  const TargetGrid& getTargetCells() {
    ndt_grid_info gi;
    grid_.leaves_.clear();
    grid_.min_points_ = prm_.min_points_per_voxel < 3 ? 3 : prm_.min_points_per_voxel;
    if (ndt_get_grid_info(h_, &gi) == NDT_OK && gi.n_leaves > 0) {
      std::vector<ndt_leaf> buf((size_t)gi.n_leaves);
      int64_t n = ndt_export_leaves(h_, buf.data(), buf.size());
      for (int64_t i = 0; i < n; ++i) grid_.leaves_.push_back({(size_t)buf[i].index, TargetGrid::Leaf{buf[i]}});
    }
    return grid_;
  }

  // ---- device-resident keyframe archive (ref: run/pipeline_ligo_tc.cpp:519-529, run/pipeline.cpp:554-557,784) ----
  // not part of pclomp: the body-frame scans stay in HBM, the sliding-window target is assembled
  // there from ids + poses (column-major 4x4 doubles, e.g. gtsam::Pose3::matrix().data())
  template <class Cloud>
  void putKeyframe(int64_t id, const Cloud& cloud) {
    if (!h_) { status_ = NDT_ERR_NO_DEVICE; return; }
    status_ = cloud.points.empty() ? ndt_keyframe_put(h_, id, nullptr, 0, 12)
                                   : ndt_keyframe_put(h_, id, &cloud.points[0].x, cloud.points.size(), sizeof(cloud.points[0]));
  }
  void eraseKeyframe(int64_t id) { status_ = h_ ? ndt_keyframe_erase(h_, id) : NDT_ERR_NO_DEVICE; }
  int64_t keyframeCount() const { return h_ ? ndt_keyframe_count(h_) : 0; }
  // poses_colmajor: ids.size() x 16 doubles
  void setInputTargetFromKeyframes(const std::vector<int64_t>& ids, const double* poses_colmajor) {
    status_ = h_ ? ndt_set_target_from_keyframes(h_, ids.data(), poses_colmajor, (int)ids.size()) : NDT_ERR_NO_DEVICE;
  }
  void setInputSourceFromKeyframe(int64_t id) {
    status_ = h_ ? ndt_set_source_from_keyframe(h_, id) : NDT_ERR_NO_DEVICE;
    if (status_ == NDT_OK) { source_.reset(); n_src_ = 0; }  // align()'s output cloud is not filled on this path
  }

  int lastStatus() const { return status_; }
  std::string lastError() const { return h_ ? ndt_last_error(h_) : "no engine (ndt_create failed: GPU required)"; }
  const ndt_result& rawResult() const { return res_; }
  ndt_handle* handle() { return h_; }

 private:
  void push() { if (h_) status_ = ndt_set_params(h_, &prm_); }
  template <class Cloud>
  void uploadTarget(const Cloud* c) {
    if (!h_) return;
    status_ = (c && !c->points.empty())
                  ? ndt_set_target(h_, &c->points[0].x, c->points.size(), sizeof(c->points[0]))
                  : ndt_set_target(h_, nullptr, 0, 12);
  }
  template <class Cloud>
  void uploadSource(const Cloud* c) {
    if (!h_) return;
    n_src_ = c ? c->points.size() : 0;
    status_ = n_src_ ? ndt_set_source(h_, &c->points[0].x, n_src_, sizeof(c->points[0]))
                     : ndt_set_source(h_, nullptr, 0, 12);
  }
  void run(const float* guess) {
    std::memset(&res_, 0, sizeof(res_));
    std::memcpy(res_.final_transformation, guess, sizeof(float) * 16);
    if (!h_) { status_ = NDT_ERR_NO_DEVICE; return; }
    status_ = ndt_align(h_, guess, &res_);
    if (status_ != NDT_OK) {  // the reference returns the prior, not converged
      std::memcpy(res_.final_transformation, guess, sizeof(float) * 16);
      res_.converged = 0;
    }
  }
  // `output` of align(): the source transformed by the result.  The reference's drivers never
  // read it (ref: run/pipeline.cpp:552,561), so it is filled on the device only when asked.
  void fillOutput(PointCloudSource& output) {
    if (!fill_output_ || !h_ || n_src_ == 0) return;
    std::vector<float> xyz(3 * n_src_);
    if (ndt_transform_source(h_, res_.final_transformation, xyz.data(), n_src_) != NDT_OK) return;
    output.points.resize(n_src_);
    for (size_t i = 0; i < n_src_; ++i) {
      output.points[i].x = xyz[3 * i];
      output.points[i].y = xyz[3 * i + 1];
      output.points[i].z = xyz[3 * i + 2];
    }
  }

 public:
  void setFillOutputCloud(bool on) { fill_output_ = on; }

 private:
  ndt_params prm_{};
  ndt_handle* h_ = nullptr;
  ndt_result res_{};
  int status_ = NDT_OK;
  size_t n_src_ = 0;
  bool fill_output_ = false;
  TargetGrid grid_;
#if !NDT_HIP_WITH_PCL
  typename PointCloudSource::ConstPtr source_;
#endif
};

// ---------------------------------------------------------------------------------------------
// svn_ndt::SvnNormalDistributionsTransform-shaped adapter (ref: extern/svn_ndt/include/svn_ndt.h:
// 100-182; driver run/pipeline_lo_svn.cpp:301-319,387-388).  Poses cross as 4x4 double,
// column-major (gtsam::Pose3::matrix().data()); build a gtsam::Pose3 from SvnNdtResult::final_pose
// on the caller's side.  Stage 1 of every SVN iteration is one batched kernel launch.
struct SvnNdtResult {
  std::array<double, 16> final_pose{};        // column-major 4x4
  std::array<double, 36> final_covariance{};  // row-major, GTSAM tangent order [rot, trans]
  bool converged = false;
  int iterations = 0;
};

template <typename PointSource, typename PointTarget>
class SvnNormalDistributionsTransform {
 public:
  SvnNormalDistributionsTransform() {
    ndt_default_params(&prm_);
    prm_.hessian_mode = NDT_HESSIAN_GAUSS_NEWTON;  // svn_ndt.h:314
    prm_.add_ridge = 1;                            // svn_ndt_impl.hpp:650-653
    status_ = ndt_create(&prm_, &h_);
    ndt_svn_default_params(&svn_);
  }
  ~SvnNormalDistributionsTransform() { ndt_destroy(h_); }
  SvnNormalDistributionsTransform(const SvnNormalDistributionsTransform&) = delete;
  SvnNormalDistributionsTransform& operator=(const SvnNormalDistributionsTransform&) = delete;

  void setResolution(float r) { prm_.resolution = r; push(); }
  void setMinPointPerVoxel(int n) { prm_.min_points_per_voxel = n; push(); }
  void setOutlierRatio(double o) { prm_.outlier_ratio = o; push(); }
  void setNeighborhoodSearchMethod(NeighborSearchMethod m) { prm_.search_method = (int)m; push(); }
  void setUseGaussNewtonHessian(bool on) { prm_.hessian_mode = on ? NDT_HESSIAN_GAUSS_NEWTON : NDT_HESSIAN_FULL; push(); }
  void setNumThreads(int n) { prm_.num_threads = n; push(); }
  void setParticleCount(int k) { svn_.particle_count = k; }
  void setMaxIterations(int n) { svn_.max_iterations = n; }
  void setKernelBandwidth(double h) { svn_.kernel_bandwidth = h; }
  void setStepSize(double s) { svn_.step_size = s; }
  void setEarlyStopThreshold(double t) { svn_.stop_threshold = t; }
  void setParticleSeed(uint64_t seed) { seed_ = seed; }  // the reference seeds from the wall clock

  template <class CloudPtr>
  void setInputTarget(const CloudPtr& cloud) {
    if (!h_) return;
    status_ = (cloud && !cloud->points.empty())
                  ? ndt_set_target(h_, &cloud->points[0].x, cloud->points.size(), sizeof(cloud->points[0]))
                  : ndt_set_target(h_, nullptr, 0, 12);
  }

  // align(source_cloud, prior_mean): prior_pose = gtsam::Pose3::matrix().data()
  template <class Cloud>
  SvnNdtResult align(const Cloud& source, const double prior_pose_colmajor[16]) {
    SvnNdtResult r;
    std::memcpy(r.final_pose.data(), prior_pose_colmajor, sizeof(double) * 16);
    for (int i = 0; i < 6; ++i) r.final_covariance[7 * i] = 1.0;  // failure convention, ref :682-702
    if (!h_) { status_ = NDT_ERR_NO_DEVICE; return r; }
    status_ = source.points.empty() ? ndt_set_source(h_, nullptr, 0, 12)
                                    : ndt_set_source(h_, &source.points[0].x, source.points.size(), sizeof(source.points[0]));
    if (status_ != NDT_OK || svn_.particle_count <= 0) return r;
    std::vector<double> particles(16 * (size_t)svn_.particle_count);
    ndt_svn_sample_particles(prior_pose_colmajor, svn_.particle_count, seed_++, particles.data());
    ndt_svn_result out;
    status_ = ndt_svn_align(h_, &svn_, prior_pose_colmajor, particles.data(), &out);
    if (status_ != NDT_OK) return r;
    std::memcpy(r.final_pose.data(), out.final_pose, sizeof(double) * 16);
    std::memcpy(r.final_covariance.data(), out.final_covariance, sizeof(double) * 36);
    r.converged = out.converged != 0;
    r.iterations = out.iterations;
    return r;
  }

  int lastStatus() const { return status_; }
  std::string lastError() const { return h_ ? ndt_last_error(h_) : "no engine (ndt_create failed: GPU required)"; }

 private:
  void push() { if (h_) status_ = ndt_set_params(h_, &prm_); }
  ndt_params prm_{};
  ndt_svn_params svn_{};
  ndt_handle* h_ = nullptr;
  int status_ = NDT_OK;
  uint64_t seed_ = 1;
};

}  // namespace ndt_hip
