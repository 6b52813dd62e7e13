// ndt_hip.hpp -- header-only C++ adapter over the C-ABI (include/ndt_hip.h) with the method
// names AND the value types of `pclomp::NormalDistributionsTransform` /
// `svn_ndt::SvnNormalDistributionsTransform` that the reference's drivers use
// (ref: run/pipeline.cpp:464-481,557-604; run/pipeline_ligo_tc.cpp:287-305,529-541;
//  run/pipeline_ins_map_distribution.cpp:346-371; run/pipeline_lo_svn.cpp:299-320,387-388;
//  include/pipeline.hpp:163-222; extern/svn_ndt/test/test_svn_ndt.cpp:144-179).
//
// Three optional faces, detected with __has_include (or forced with -DNDT_HIP_WITH_*=0/1):
//   * Eigen  (<Eigen/Core>): Matrix4f / Matrix6d / Vector3d / Matrix3d ARE the Eigen types, so
//     `ndt_result.hessian + Matrix6d::Identity() * 1e-6` (run/pipeline.cpp:594-596) and
//     `.mean = leaf.getMean()` (include/pipeline.hpp:196-201) compile as written.
//     Without Eigen they are small column-major POD matrices with the same accessors.
//   * PCL    (<pcl/registration/registration.h>, needs Eigen): the engine derives from
//     `pcl::Registration`, so it can be stored in `RegisterCallback::registration`
//     (ref: include/registercallback.hpp:35) and driven through setInputTarget /
//     setInputSource / align / getFinalTransformation / hasConverged.
//   * GTSAM  (<gtsam/geometry/Pose3.h>): `SvnNormalDistributionsTransform::align(cloud,
//     gtsam::Pose3)` returning `SvnNdtResult{gtsam::Pose3 final_pose; Matrix6d
//     final_covariance; ...}` (ref: extern/svn_ndt/include/svn_ndt.h:40-51,100-182).
// include/compat/ holds `pclomp/ndt_omp.h`, `svn_ndt.h`, ... that alias the reference's
// namespaces onto this header: with that directory in front of extern/ndt_omp/include and
// extern/svn_ndt/include the drivers compile unchanged.
//
// The build image has neither Eigen, PCL nor GTSAM: the three faces are compile- and run-tested
// against minimal API mocks (tests/cpp/mock/, test doubles -- not the libraries), the plain face
// by tests/cpp/test_adapter.cpp.
//
// Failures never throw: like the reference (ref: svn_ndt_impl.hpp:682-702) a failed align
// leaves the guess as the final transformation and hasConverged() == false; the status and
// message are available through lastStatus() / lastError().
#pragma once

#include <algorithm>
#include <cmath>
#include <array>
#include <cstddef>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "../ndt_hip.h"

#if !defined(NDT_HIP_WITH_EIGEN)
#if defined(__has_include)
#if __has_include(<Eigen/Core>)
#define NDT_HIP_WITH_EIGEN 1
#endif
#endif
#endif
#ifndef NDT_HIP_WITH_EIGEN
#define NDT_HIP_WITH_EIGEN 0
#endif

#if !defined(NDT_HIP_WITH_PCL)
#if NDT_HIP_WITH_EIGEN && defined(__has_include)
#if __has_include(<pcl/registration/registration.h>)
#define NDT_HIP_WITH_PCL 1
#endif
#endif
#endif
#ifndef NDT_HIP_WITH_PCL
#define NDT_HIP_WITH_PCL 0
#endif

#if !defined(NDT_HIP_WITH_GTSAM)
#if NDT_HIP_WITH_EIGEN && defined(__has_include)
#if __has_include(<gtsam/geometry/Pose3.h>)
#define NDT_HIP_WITH_GTSAM 1
#endif
#endif
#endif
#ifndef NDT_HIP_WITH_GTSAM
#define NDT_HIP_WITH_GTSAM 0
#endif

#if NDT_HIP_WITH_EIGEN
#include <Eigen/Core>
#endif
#if NDT_HIP_WITH_PCL
#include <pcl/point_cloud.h>
#include <pcl/registration/registration.h>
#endif
#if NDT_HIP_WITH_GTSAM
#include <gtsam/geometry/Pose3.h>
#endif

namespace ndt_hip {

// same order as pclomp::NeighborSearchMethod (ref: run/pipeline.cpp:471-480)
enum NeighborSearchMethod { KDTREE = NDT_KDTREE, DIRECT26 = NDT_DIRECT26, DIRECT7 = NDT_DIRECT7, DIRECT1 = NDT_DIRECT1 };
// svn_ndt::NeighborSearchMethod (ref: extern/svn_ndt/include/svn_ndt.h:30-35)
enum class SvnNeighborSearchMethod { KDTREE, DIRECT7, DIRECT1 };

// ---- value types ----------------------------------------------------------------------------
#if NDT_HIP_WITH_EIGEN
using Matrix4f = Eigen::Matrix<float, 4, 4>;
using Matrix4d = Eigen::Matrix<double, 4, 4>;
using Matrix6d = Eigen::Matrix<double, 6, 6>;
using Matrix3d = Eigen::Matrix<double, 3, 3>;
using Vector3d = Eigen::Matrix<double, 3, 1>;
#else
// column-major like Eigen's default, same accessors: data(), (r, c), [i]
template <typename T, int R, int C>
struct Mat {
  T m[R * C];
  T* data() { return m; }
  const T* data() const { return m; }
  T& operator()(int r, int c) { return m[c * R + r]; }
  const T& operator()(int r, int c) const { return m[c * R + r]; }
  T& operator[](int i) { return m[i]; }
  const T& operator[](int i) const { return m[i]; }
  bool operator==(const Mat& o) const { return std::equal(m, m + R * C, o.m); }
  bool operator!=(const Mat& o) const { return !(*this == o); }
};
using Matrix4f = Mat<float, 4, 4>;
using Matrix4d = Mat<double, 4, 4>;
using Matrix6d = Mat<double, 6, 6>;
using Matrix3d = Mat<double, 3, 3>;
using Vector3d = Mat<double, 3, 1>;
#endif

namespace detail {
template <class M>
inline M from_rowmajor(const double* a, int rows, int cols) {
  M out;
  for (int r = 0; r < rows; ++r)
    for (int c = 0; c < cols; ++c) out(r, c) = a[cols * r + c];
  return out;
}
template <class M>
inline void to_rowmajor(const M& in, int rows, int cols, double* a) {
  for (int r = 0; r < rows; ++r)
    for (int c = 0; c < cols; ++c) a[cols * r + c] = in(r, c);
}
template <class M, class S>
inline M from_colmajor(const S* a, int rows, int cols) {
  M out;
  for (int c = 0; c < cols; ++c)
    for (int r = 0; r < rows; ++r) out(r, c) = a[rows * c + r];
  return out;
}
template <class M, class S>
inline void to_colmajor(const M& in, int rows, int cols, S* a) {
  for (int c = 0; c < cols; ++c)
    for (int r = 0; r < rows; ++r) a[rows * c + r] = (S)in(r, c);
}
}  // namespace detail

inline Matrix4f identity4f() {
  const float I[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  return detail::from_colmajor<Matrix4f>(I, 4, 4);
}

// pclomp::NdtResult (fields the drivers read: iteration_num, hessian -- run/pipeline.cpp:567-568,
// 594; run/pipeline_ligo_tc.cpp:536-538)
struct NdtResult {
  Matrix4f pose = identity4f();
  float transform_probability = 0.0f;
  float nearest_voxel_transformation_likelihood = 0.0f;
  int iteration_num = 0;
  // [RECALLED: tier4 ndt_omp] entry 0 = the initial guess, then one entry per iteration (no driver reads them)
  std::vector<Matrix4f> transformation_array;
  std::vector<float> transform_probability_array;
  std::vector<float> nearest_voxel_transformation_likelihood_array;
  Matrix6d hessian{};  // of the maximised score: the drivers form cov = -(hessian + 1e-6 I)^-1
  // that covariance, optionally in the block order the drivers hand to GTSAM
  // (ref: run/pipeline.cpp:594-603, src/registercallback.cpp:170-186); false if singular
  bool covarianceForGtsam(Matrix6d& cov, double eps = 1e-6, bool gtsam_order = true) const {
    double h[36], c[36];
    detail::to_rowmajor(hessian, 6, 6, h);
    if (ndt_result_covariance(h, eps, gtsam_order ? 1 : 0, c) != NDT_OK) return false;
    cov = detail::from_rowmajor<Matrix6d>(c, 6, 6);
    return true;
  }
};

// getTargetCells(): what extractNdtData() uses (ref: include/pipeline.hpp:175-206), i.e.
// pclomp::VoxelGridCovariance<PointT>: getLeaves() iterable as {index, leaf} in ascending index
// order, getMinPointPerVoxel(), getLeafCenter(index); Leaf accessors returning Eigen values.
class TargetGrid {
 public:
  struct Leaf {
    ndt_leaf d;
    int getPointCount() const { return d.point_count; }
    Vector3d getMean() const { return detail::from_colmajor<Vector3d>(d.mean, 3, 1); }
    Matrix3d getCov() const { return detail::from_rowmajor<Matrix3d>(d.cov, 3, 3); }
    Matrix3d getInverseCov() const { return detail::from_rowmajor<Matrix3d>(d.icov, 3, 3); }
    Matrix3d getEvecs() const { return detail::from_rowmajor<Matrix3d>(d.evecs, 3, 3); }  // columns = eigenvectors
    Vector3d getEvals() const { return detail::from_colmajor<Vector3d>(d.evals, 3, 1); }
  };
  using Entry = std::pair<size_t, Leaf>;  // {voxel index, leaf}
  const std::vector<Entry>& getLeaves() const { return leaves_; }
  int getMinPointPerVoxel() const { return min_points_; }
  // O(log V): the leaves are kept in ascending index order.  Like the reference's getLeaf(index), only a VALID leaf is
  // handed out -- at least getMinPointPerVoxel() points (a leaf whose covariance could not be inverted carries -1; ref:
  // voxel_grid_covariance.h:262-278).  (The engine exports the valid leaves only, so getLeaves() iterates over those; the
  // reference's map also holds the voxels with too few points, which its users filter out: include/pipeline.hpp:186.)
  const Leaf* getLeaf(size_t index) const {
    auto it = std::lower_bound(leaves_.begin(), leaves_.end(), index,
                               [](const Entry& e, size_t i) { return e.first < i; });
    return (it != leaves_.end() && it->first == index && it->second.d.point_count >= min_points_) ? &it->second : nullptr;
  }
  Vector3d getLeafCenter(size_t index) const {
    double c[3] = {0.0, 0.0, 0.0};
    if (const Leaf* l = getLeaf(index))
      for (int a = 0; a < 3; ++a) c[a] = (double)l->d.center[a];
    return detail::from_colmajor<Vector3d>(c, 3, 1);
  }

  // ---- the reference grid's query surface (round 5; ref: extern/svn_ndt/include/voxel_grid_covariance.h:194-200,280-381,
  // voxel_grid_covariance_impl.hpp:46-71,455-615), answered on the host from the exported leaves and the grid geometry --
  // the same f32 bounds test and index arithmetic, so a query lands in the voxel the engine's kernels put it in.  Points
  // are anything with x / y / z members (pcl::PointXYZ, PointXYZI, the PCL-free point types of this header) or
  // operator[] (Eigen::Vector3f).  No driver calls these; they exist so that code written against the reference's grid
  // class keeps compiling.
  using LeafConstPtr = const Leaf*;
  double getCovEigValueInflationRatio() const { return eig_ratio_; }
  bool isPointWithinBounds(float px, float py, float pz) const {   // ref: voxel_grid_covariance_impl.hpp:46-71
    if (gi_.inverse_leaf_size == 0.0f) return false;
    const float w = gi_.leaf_size;
    return px >= (float)gi_.min_b[0] * w && px < (float)(gi_.max_b[0] + 1) * w && py >= (float)gi_.min_b[1] * w &&
           py < (float)(gi_.max_b[1] + 1) * w && pz >= (float)gi_.min_b[2] * w && pz < (float)(gi_.max_b[2] + 1) * w;
  }
  LeafConstPtr getLeafAt(float px, float py, float pz) const {    // ref: voxel_grid_covariance.h:280-303
    if (!isPointWithinBounds(px, py, pz)) return nullptr;
    const float il = gi_.inverse_leaf_size;
    const int i0 = static_cast<int>(std::floor(px * il) - (float)gi_.min_b[0]);
    const int i1 = static_cast<int>(std::floor(py * il) - (float)gi_.min_b[1]);
    const int i2 = static_cast<int>(std::floor(pz * il) - (float)gi_.min_b[2]);
    return getLeaf(static_cast<size_t>(i0 + i1 * gi_.div_b[0] + i2 * gi_.div_b[0] * gi_.div_b[1]));
  }
  template <class P>
  auto getLeaf(const P& p) const -> decltype((void)p.x, LeafConstPtr()) { return getLeafAt(p.x, p.y, p.z); }
  template <class V>
  auto getLeaf(const V& v) const -> decltype((void)v[0], (void)v.data(), LeafConstPtr()) { return getLeafAt((float)v[0], (float)v[1], (float)v[2]); }
  // DIRECT7 / DIRECT1 (ref: voxel_grid_covariance_impl.hpp:560-615): the point offset by +-leaf in f32 and re-classified
  template <class P>
  int getNeighborhoodAtPoint7(const P& ref, std::vector<LeafConstPtr>& neighbors) const {
    neighbors.clear();
    neighbors.reserve(7);
    const float w = gi_.leaf_size, x = ref.x, y = ref.y, z = ref.z;
    if (LeafConstPtr c = getLeafAt(x, y, z)) neighbors.push_back(c);
    if (!(w > 0.0f)) return (int)neighbors.size();
    const float probe[6][3] = {{x + w, y, z}, {x - w, y, z}, {x, y + w, z}, {x, y - w, z}, {x, y, z + w}, {x, y, z - w}};
    for (const auto& q : probe)
      if (LeafConstPtr l = getLeafAt(q[0], q[1], q[2])) neighbors.push_back(l);
    return (int)neighbors.size();
  }
  template <class P>
  int getNeighborhoodAtPoint1(const P& ref, std::vector<LeafConstPtr>& neighbors) const {
    neighbors.clear();
    if (LeafConstPtr l = getLeafAt(ref.x, ref.y, ref.z)) neighbors.push_back(l);
    return (int)neighbors.size();
  }
  // the centroid cloud the reference's kd-tree is built on: the f32-rounded means of the valid leaves, ascending voxel
  // index (ref: voxel_grid_covariance_impl.hpp:400-440)
  struct Centroid { float x, y, z; };
  const std::vector<Centroid>& getCentroids() const {
    if (!centroids_built_) {
      centroids_.clear();
      centroid_leaf_.clear();
      for (size_t i = 0; i < leaves_.size(); ++i) {
        const ndt_leaf& d = leaves_[i].second.d;
        if (d.point_count < min_points_) continue;
        centroids_.push_back(Centroid{(float)d.mean[0], (float)d.mean[1], (float)d.mean[2]});
        centroid_leaf_.push_back(i);
      }
      centroids_built_ = true;
    }
    return centroids_;
  }
  // FLANN's L2_Simple in f32 (accumulated x, y, z), strict `<` on the squared radius, results by ascending distance -- what
  // pcl::KdTreeFLANN::radiusSearch returns (ref: voxel_grid_covariance_impl.hpp:505-554).  A linear scan of the centroids: a
  // host-side convenience, not the engine's KDTREE path (that one is the 27-cell scan inside k_derivatives).
  template <class P>
  int radiusSearch(const P& point, double radius, std::vector<LeafConstPtr>& k_leaves, std::vector<float>& k_sqr_distances,
                   unsigned int max_nn = 0) const {
    k_leaves.clear();
    k_sqr_distances.clear();
    if (radius <= 0.0) return 0;
    const float r2 = (float)(radius * radius);
    std::vector<std::pair<float, size_t>> hits;
    const std::vector<Centroid>& c = getCentroids();
    for (size_t i = 0; i < c.size(); ++i) {
      const float d = sqr_distance(c[i], point.x, point.y, point.z);
      if (d < r2) hits.emplace_back(d, i);
    }
    std::sort(hits.begin(), hits.end());
    if (max_nn > 0 && hits.size() > max_nn) hits.resize(max_nn);
    for (const auto& h : hits) { k_leaves.push_back(&leaves_[centroid_leaf_[h.second]].second); k_sqr_distances.push_back(h.first); }
    return (int)k_leaves.size();
  }
  template <class P>
  int nearestKSearch(const P& point, int k, std::vector<LeafConstPtr>& k_leaves, std::vector<float>& k_sqr_distances) const {
    k_leaves.clear();
    k_sqr_distances.clear();
    if (k <= 0) return 0;
    std::vector<std::pair<float, size_t>> all;
    const std::vector<Centroid>& c = getCentroids();
    all.reserve(c.size());
    for (size_t i = 0; i < c.size(); ++i) all.emplace_back(sqr_distance(c[i], point.x, point.y, point.z), i);
    const size_t n = std::min(all.size(), (size_t)k);
    std::partial_sort(all.begin(), all.begin() + n, all.end());
    for (size_t i = 0; i < n; ++i) { k_leaves.push_back(&leaves_[centroid_leaf_[all[i].second]].second); k_sqr_distances.push_back(all[i].first); }
    return (int)n;
  }

  std::vector<Entry> leaves_;
  int min_points_ = 6;
  ndt_grid_info gi_{};          // geometry of the grid the leaves were exported from
  double eig_ratio_ = 0.01;

 private:
  static float sqr_distance(const Centroid& c, float px, float py, float pz) {
    const float ex = px - c.x, ey = py - c.y, ez = pz - c.z;
    float d = ex * ex;
    d = d + ey * ey;
    d = d + ez * ez;
    return d;
  }
  mutable std::vector<Centroid> centroids_;
  mutable std::vector<size_t> centroid_leaf_;   // centroid -> position in leaves_
  mutable bool centroids_built_ = false;
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
#if NDT_HIP_WITH_PCL
  // whatever smart pointer this PCL uses (std:: since 1.11, boost:: before), so that
  // `registerCallback.registration = ndt_omp;` (ref: run/pipeline.cpp:481) converts
  using Base = pcl::Registration<PointSource, PointTarget>;
  using Ptr = typename std::pointer_traits<typename Base::Ptr>::template rebind<NormalDistributionsTransform<PointSource, PointTarget>>;
  using ConstPtr = typename std::pointer_traits<typename Base::Ptr>::template rebind<const NormalDistributionsTransform<PointSource, PointTarget>>;
#else
  using Ptr = std::shared_ptr<NormalDistributionsTransform<PointSource, PointTarget>>;
  using ConstPtr = std::shared_ptr<const NormalDistributionsTransform<PointSource, PointTarget>>;
#endif

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
  NeighborSearchMethod getNeighborhoodSearchMethod() const { return (NeighborSearchMethod)prm_.search_method; }
  void setRegularizationScaleFactor(float k) { prm_.regularization_scale_factor = k; push(); }
  void setMinPointPerVoxel(int n) { prm_.min_points_per_voxel = n; push(); }
  void setTransformationEpsilon(double e) {
#if NDT_HIP_WITH_PCL
    this->transformation_epsilon_ = e;
#endif
    prm_.trans_epsilon = e;
    push();
  }
  void setMaximumIterations(int n) {
#if NDT_HIP_WITH_PCL
    this->max_iterations_ = n;
#endif
    prm_.max_iterations = n;
    push();
  }
  void setRegularizationPose(const Matrix4f& T) {  // ref: run/pipeline_ligo_tc.cpp:531
    float a[16];
    detail::to_colmajor(T, 4, 4, a);
    status_ = h_ ? ndt_set_regularization_pose(h_, a) : NDT_ERR_NO_DEVICE;
  }
  void unsetRegularizationPose() { status_ = h_ ? ndt_clear_regularization_pose(h_) : NDT_ERR_NO_DEVICE; }
  // every engine parameter at once (hessian_mode, cov_mode, add_ridge, use_line_search, ...)
  const ndt_params& params() const { return prm_; }
  void setParams(const ndt_params& p) { const int dev = prm_.device_id; prm_ = p; prm_.device_id = dev; push(); }

  // ---- clouds ----
#if NDT_HIP_WITH_PCL
  void setInputTarget(const typename Base::PointCloudTargetConstPtr& cloud) override {
    Base::setInputTarget(cloud);
    // pcl::Registration::initCompute() would build a FLANN kd-tree over the whole target on the
    // next align() (hundreds of ms for a 1M-point map); NDT never queries it
    this->target_cloud_updated_ = false;
    uploadTarget(cloud.get());
  }
  void setInputSource(const typename Base::PointCloudSourceConstPtr& cloud) override {
    Base::setInputSource(cloud);
    this->source_cloud_updated_ = false;
    uploadSource(cloud.get());
  }
#else
  void setInputTarget(const typename PointCloudTarget::ConstPtr& cloud) { uploadTarget(cloud.get()); }
  void setInputSource(const typename PointCloudSource::ConstPtr& cloud) { uploadSource(cloud.get()); }
#endif

  // ---- registration ----
  // public in pclomp too (ref: extern/svn_ndt/test/test_svn_ndt.cpp:171); called by
  // pcl::Registration::align(output, guess) in the PCL face
  void computeTransformation(PointCloudSource& output, const Matrix4f& guess)
#if NDT_HIP_WITH_PCL
      override
#endif
  {
    float g[16];
    detail::to_colmajor(guess, 4, 4, g);
    run(g);
#if NDT_HIP_WITH_PCL
    this->final_transformation_ = detail::from_colmajor<Matrix4f>(res_.final_transformation, 4, 4);
    this->transformation_ = this->final_transformation_;
    this->converged_ = res_.converged != 0;
    this->nr_iterations_ = res_.iterations;
#endif
    fillOutput(output);
  }
#if !NDT_HIP_WITH_PCL
  void align(PointCloudSource& output, const Matrix4f& guess = identity4f()) { computeTransformation(output, guess); }
  Matrix4f getFinalTransformation() const { return detail::from_colmajor<Matrix4f>(res_.final_transformation, 4, 4); }
  bool hasConverged() const { return res_.converged != 0; }
#endif
  int getFinalNumIteration() const { return res_.iterations; }
  double getTransformationProbability() const { return res_.transform_probability; }
  double getNearestVoxelTransformationLikelihood() const { return res_.nearest_voxel_transformation_likelihood; }

  // scoring-only calls of pclomp: score of `cloud` under transform T against the
  // current target, no gradient, nothing about the engine's source / result changes except the
  // source cloud (replaced by `cloud`, as pclomp's versions take the cloud to score)
  template <class Cloud>
  double calculateTransformationProbability(const Cloud& cloud, const Matrix4f& T = identity4f()) {
    ndt_score s;
    return scoreCloud(cloud, T, &s) ? s.transform_probability : 0.0;
  }
  template <class Cloud>
  double calculateNearestVoxelTransformationLikelihood(const Cloud& cloud, const Matrix4f& T = identity4f()) {
    ndt_score s;
    return scoreCloud(cloud, T, &s) ? s.nearest_voxel_transformation_likelihood : 0.0;
  }

  NdtResult getResult() const {
    NdtResult r;
    r.pose = detail::from_colmajor<Matrix4f>(res_.final_transformation, 4, 4);
    r.transform_probability = (float)res_.transform_probability;
    r.nearest_voxel_transformation_likelihood = (float)res_.nearest_voxel_transformation_likelihood;
    r.iteration_num = res_.iterations;
    r.hessian = detail::from_rowmajor<Matrix6d>(res_.hessian, 6, 6);
    const int n = h_ ? ndt_get_iteration_history(h_, nullptr, nullptr, nullptr, 0) : 0;
    if (n > 0) {
      std::vector<float> T((size_t)n * 16);
      std::vector<double> tp((size_t)n), nv((size_t)n);
      ndt_get_iteration_history(h_, T.data(), tp.data(), nv.data(), n);
      for (int i = 0; i < n; ++i) {
        r.transformation_array.push_back(detail::from_colmajor<Matrix4f>(T.data() + 16 * (size_t)i, 4, 4));
        r.transform_probability_array.push_back((float)tp[(size_t)i]);
        r.nearest_voxel_transformation_likelihood_array.push_back((float)nv[(size_t)i]);
      }
    }
    return r;
  }

  // ---- voxel grid (ref: include/pipeline.hpp:178-206) ----
  const TargetGrid& getTargetCells() {
    ndt_grid_info gi;
    grid_ = TargetGrid();
    grid_.min_points_ = prm_.min_points_per_voxel < 3 ? 3 : prm_.min_points_per_voxel;
    grid_.eig_ratio_ = prm_.eig_inflation_ratio;
    if (h_ && ndt_get_grid_info(h_, &gi) == NDT_OK && gi.n_leaves > 0) {
      grid_.gi_ = gi;
      std::vector<ndt_leaf> buf((size_t)gi.n_leaves);
      const int64_t n = ndt_export_leaves(h_, buf.data(), buf.size());  // ascending index
      grid_.leaves_.reserve(n > 0 ? (size_t)n : 0);
      for (int64_t i = 0; i < n; ++i) grid_.leaves_.emplace_back((size_t)buf[i].index, TargetGrid::Leaf{buf[i]});
    }
    return grid_;
  }

  // ---- pcl::VoxelGrid on the device (ref: run/pipeline_ins_map_distribution.cpp:324-340: vg.setLeafSize(vs, vs, vs);
  // vg.setInputCloud(map); vg.filter(*ds_map)) ----
  // `out` receives the centroid (x, y, z and, for point types with an `intensity` member at byte 16, the intensity) of
  // every occupied voxel in ascending voxel index; the engine's target and source are left as they are
  template <class P>
  static auto intensity_offset_of(int) -> decltype((void)static_cast<const float*>(&std::declval<const P&>().intensity), int()) {
    static const P probe{};
    return (int)(reinterpret_cast<const char*>(&probe.intensity) - reinterpret_cast<const char*>(&probe.x));
  }
  template <class P>
  static int intensity_offset_of(long) { return -1; }

  template <class Cloud>
  void voxelDownsample(const Cloud& in, float leaf, Cloud& out) {
    out.points.clear();
    if (!h_) { status_ = NDT_ERR_NO_DEVICE; return; }
    if (in.points.empty()) { status_ = NDT_OK; return; }
    using P = typename std::decay<decltype(in.points[0])>::type;
    out.points.resize(in.points.size());   // value-initialised points: the fields the engine does not write keep their defaults
    size_t m = 0;
    // the intensity is averaged only for point types that HAVE a float member of that name, at ITS offset (pcl::PointXYZRGB
    // and PointNormal are 32 bytes too and carry packed colour / a normal at byte 16: never averaged as a float)
    status_ = ndt_voxel_downsample(h_, &in.points[0].x, in.points.size(), sizeof(P), intensity_offset_of<P>(0), leaf,
                                   &out.points[0].x, out.points.size(), &m);
    out.points.resize(status_ == NDT_OK ? m : 0);
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
    if (status_ == NDT_OK) n_src_ = 0;  // align()'s output cloud is not filled on this path
  }

  // ---- clouds that are already in HBM (SoA float arrays on this engine's device) ----
  // the target arrays are consumed by the build; the source arrays are viewed in place until the source
  // is replaced (setInputSource's shared_ptr contract) -- or copied with copy = true
  // (deferred = true: the build is only enqueued -- the arrays must stay unchanged until align() / wait() has returned --
  // and the align's first evaluation goes onto the stream behind it: ndt_set_target_device_deferred)
  void setInputTargetDevice(const float* dx, const float* dy, const float* dz, size_t n, bool deferred = false) {
    if (!h_) { status_ = NDT_ERR_NO_DEVICE; return; }
    status_ = deferred ? ndt_set_target_device_deferred(h_, dx, dy, dz, n) : ndt_set_target_device(h_, dx, dy, dz, n);
  }
  void setInputSourceDevice(const float* dx, const float* dy, const float* dz, size_t n, bool copy = false) {
    if (!h_) { status_ = NDT_ERR_NO_DEVICE; return; }
    status_ = copy ? ndt_set_source_device(h_, dx, dy, dz, n) : ndt_set_source_device_view(h_, dx, dy, dz, n);
    if (status_ == NDT_OK) n_src_ = 0;  // align()'s output cloud is not filled on this path
  }

  // engine-specific: the hand-off of host clouds.  By default setInputTarget / setInputSource return as soon as the
  // caller's cloud has been consumed; the transfer and the voxel-grid build finish behind them and align() waits
  // (ndt_set_handoff_mode in ndt_hip.h).  wait() blocks until they are complete and reports a failed build.
  void setHandoffMode(int mode /* NDT_HANDOFF_ASYNC | NDT_HANDOFF_SYNC */) {
    status_ = h_ ? ndt_set_handoff_mode(h_, mode) : NDT_ERR_NO_DEVICE;
  }
  int wait() { return status_ = h_ ? ndt_wait(h_) : NDT_ERR_NO_DEVICE; }

  // engine-specific: 48-byte voxel records (f64 mean, f32 inverse covariance) instead of 80-byte f64 ones
  // (ndt_set_record_format in ndt_hip.h: what it costs in accuracy and what it saves per evaluation)
  void setPackedVoxelRecords(bool on) {
    status_ = h_ ? ndt_set_record_format(h_, on ? NDT_RECORDS_PACKED48 : NDT_RECORDS_F64) : NDT_ERR_NO_DEVICE;
  }

  // ---- multi-grid target [RECALLED: tier4 ndt_omp multigrid_ndt_omp.h -- addTarget / removeTarget /
  // createVoxelKdtree with string ids; the reference names the class only in its build,
  // CMakeLists.txt:41-42, and no driver instantiates it] ----
  template <class CloudPtrT>
  void addTarget(const CloudPtrT& cloud, const std::string& target_id) {
    if (!h_) { status_ = NDT_ERR_NO_DEVICE; return; }
    if (!cloud || cloud->points.empty()) { status_ = NDT_ERR_INVALID_ARG; return; }
    auto it = target_ids_.find(target_id);
    const int64_t id = it != target_ids_.end() ? it->second : next_target_id_++;
    status_ = ndt_multigrid_add_target(h_, id, &cloud->points[0].x, cloud->points.size(), sizeof(cloud->points[0]));
    if (status_ == NDT_OK) target_ids_[target_id] = id;
    grid_ = TargetGrid();
  }
  template <class CloudPtrT>
  void setInputTarget(const CloudPtrT& cloud, const std::string& target_id) { addTarget(cloud, target_id); }
  void removeTarget(const std::string& target_id) {
    auto it = target_ids_.find(target_id);
    if (!h_ || it == target_ids_.end()) { status_ = h_ ? NDT_ERR_INVALID_ARG : NDT_ERR_NO_DEVICE; return; }
    status_ = ndt_multigrid_remove_target(h_, it->second);
    target_ids_.erase(it);
    grid_ = TargetGrid();
  }
  void createVoxelKdtree() {
    status_ = h_ ? ndt_multigrid_create_kdtree(h_) : NDT_ERR_NO_DEVICE;
    grid_ = TargetGrid();
  }
  std::vector<std::string> getCurrentMapIDs() const {
    std::vector<std::string> v;
    for (const auto& kv : target_ids_) v.push_back(kv.first);
    return v;
  }

  int lastStatus() const { return status_; }
  std::string lastError() const { return h_ ? ndt_last_error(h_) : "no engine (ndt_create failed: GPU required)"; }
  const ndt_result& rawResult() const { return res_; }
  ndt_handle* handle() { return h_; }
  // `output` of align(): the drivers never read it (ref: run/pipeline.cpp:552,561), so it is
  // only produced (on the device) when asked for
  void setFillOutputCloud(bool on) { fill_output_ = on; }

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
  template <class Cloud>
  bool scoreCloud(const Cloud& cloud, const Matrix4f& T, ndt_score* s) {
    if (!h_) { status_ = NDT_ERR_NO_DEVICE; return false; }
    uploadSource(&cloud);
    if (status_ != NDT_OK) return false;
    float a[16];
    detail::to_colmajor(T, 4, 4, a);
    status_ = ndt_score_transform(h_, a, s);
    return status_ == NDT_OK;
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

  ndt_params prm_{};
  ndt_handle* h_ = nullptr;
  ndt_result res_{};
  int status_ = NDT_OK;
  size_t n_src_ = 0;
  bool fill_output_ = false;
  TargetGrid grid_;
  std::map<std::string, int64_t> target_ids_;  // multi-grid: tier4's string ids -> the C-ABI's integers
  int64_t next_target_id_ = 1;
};

// tier4's class name for the multi-grid face [RECALLED]; here the same engine object carries both faces
template <typename PointSource, typename PointTarget>
using MultiGridNormalDistributionsTransform = NormalDistributionsTransform<PointSource, PointTarget>;

// ---------------------------------------------------------------------------------------------
// svn_ndt::SvnNormalDistributionsTransform-shaped adapter (ref: extern/svn_ndt/include/svn_ndt.h:
// 100-182; driver run/pipeline_lo_svn.cpp:299-320,387-388).  Stage 1 of every SVN iteration is
// one batched kernel launch.
struct SvnNdtResult {  // svn_ndt::SvnNdtResult (svn_ndt.h:40-51)
#if NDT_HIP_WITH_GTSAM
  gtsam::Pose3 final_pose;
#else
  Matrix4d final_pose{};
#endif
  Matrix6d final_covariance{};  // GTSAM tangent order [rot, trans]
  bool converged = false;
  int iterations = 0;
};

template <typename PointSource, typename PointTarget>
class SvnNormalDistributionsTransform {
 public:
  using PointCloudSource = PointCloud<PointSource>;
  using PointCloudTarget = PointCloud<PointTarget>;

  SvnNormalDistributionsTransform() {
    ndt_default_params(&prm_);
    prm_.hessian_mode = NDT_HESSIAN_GAUSS_NEWTON;  // svn_ndt.h:314
    prm_.add_ridge = 1;                            // svn_ndt_impl.hpp:650-653
    status_ = ndt_create(&prm_, &h_);
    ndt_svn_default_params(&svn_);
    updateNdtConstants();
  }
  ~SvnNormalDistributionsTransform() { ndt_destroy(h_); }
  SvnNormalDistributionsTransform(const SvnNormalDistributionsTransform&) = delete;
  SvnNormalDistributionsTransform& operator=(const SvnNormalDistributionsTransform&) = delete;

  // clamps as the reference's setters (svn_ndt.h:118-160)
  void setResolution(float r) { prm_.resolution = r; push(); }
  float getResolution() const { return prm_.resolution; }
  void setMinPointPerVoxel(int n) { prm_.min_points_per_voxel = n; push(); }
  void setOutlierRatio(double o) { prm_.outlier_ratio = o; push(); }
  double getOutlierRatio() const { return prm_.outlier_ratio; }
  void setNeighborhoodSearchMethod(NeighborSearchMethod m) { prm_.search_method = (int)m; push(); }
  void setNeighborhoodSearchMethod(SvnNeighborSearchMethod m) {
    prm_.search_method = m == SvnNeighborSearchMethod::KDTREE ? NDT_KDTREE
                       : m == SvnNeighborSearchMethod::DIRECT1 ? NDT_DIRECT1 : NDT_DIRECT7;
    push();
  }
  void setUseGaussNewtonHessian(bool on) { prm_.hessian_mode = on ? NDT_HESSIAN_GAUSS_NEWTON : NDT_HESSIAN_FULL; push(); }
  void setNumThreads(int n) { prm_.num_threads = n > 0 ? n : 1; push(); }
  int getNumThreads() const { return prm_.num_threads; }
  void setParticleCount(int k) { svn_.particle_count = k > 0 ? k : 1; }
  int getParticleCount() const { return svn_.particle_count; }
  void setMaxIterations(int n) { svn_.max_iterations = n > 0 ? n : 1; }
  int getMaxIterations() const { return svn_.max_iterations; }
  void setKernelBandwidth(double h) { svn_.kernel_bandwidth = h > 1e-9 ? h : 1e-9; }
  double getKernelBandwidth() const { return svn_.kernel_bandwidth; }
  void setStepSize(double s) { svn_.step_size = s > 0 ? s : 1e-6; }
  double getStepSize() const { return svn_.step_size; }
  void setEarlyStopThreshold(double t) { svn_.stop_threshold = t >= 0 ? t : 1e-4; }
  double getEarlyStopThreshold() const { return svn_.stop_threshold; }
  void setParticleSeed(uint64_t seed) { seed_ = seed; }  // the reference seeds from the wall clock (:712)

  template <class CloudPtr>
  void setInputTarget(const CloudPtr& cloud) {
    if (!h_) return;
    status_ = (cloud && !cloud->points.empty())
                  ? ndt_set_target(h_, &cloud->points[0].x, cloud->points.size(), sizeof(cloud->points[0]))
                  : ndt_set_target(h_, nullptr, 0, 12);
  }

  // align(source_cloud, prior_mean) with the prior as 16 column-major doubles
  template <class Cloud>
  SvnNdtResult align(const Cloud& source, const double prior_pose_colmajor[16]) {
    ndt_svn_result out;
    std::memset(&out, 0, sizeof(out));
    std::memcpy(out.final_pose, prior_pose_colmajor, sizeof(double) * 16);
    for (int i = 0; i < 6; ++i) out.final_covariance[7 * i] = 1.0;  // failure convention, ref :682-702
    if (!h_) {
      status_ = NDT_ERR_NO_DEVICE;
    } else {
      status_ = source.points.empty() ? ndt_set_source(h_, nullptr, 0, 12)
                                      : ndt_set_source(h_, &source.points[0].x, source.points.size(), sizeof(source.points[0]));
      n_source_ = status_ == NDT_OK ? source.points.size() : 0;
      if (status_ == NDT_OK && svn_.particle_count > 0) {
        std::vector<double> particles(16 * (size_t)svn_.particle_count);
        ndt_svn_sample_particles(prior_pose_colmajor, svn_.particle_count, seed_++, particles.data());
        ndt_svn_result tmp;
        status_ = ndt_svn_align(h_, &svn_, prior_pose_colmajor, particles.data(), &tmp);
        if (status_ == NDT_OK) out = tmp;
      }
    }
    SvnNdtResult r;
    const Matrix4d P = detail::from_colmajor<Matrix4d>(out.final_pose, 4, 4);
#if NDT_HIP_WITH_GTSAM
    r.final_pose = gtsam::Pose3(P);
#else
    r.final_pose = P;
#endif
    r.final_covariance = detail::from_rowmajor<Matrix6d>(out.final_covariance, 6, 6);
    r.converged = out.converged != 0;
    r.iterations = out.iterations;
    return r;
  }
#if NDT_HIP_WITH_GTSAM
  // the reference's signature (svn_ndt.h:178-181; call site run/pipeline_lo_svn.cpp:388)
  template <class Cloud>
  SvnNdtResult align(const Cloud& source, const gtsam::Pose3& prior_mean) {
    double a[16];
    detail::to_colmajor(prior_mean.matrix(), 4, 4, a);
    return align(source, a);
  }
#endif

  // ---- the reference's public math hook (round 5; ref: extern/svn_ndt/include/svn_ndt.h:186-206, svn_ndt_impl.hpp:518-668) ----
  // computeParticleDerivatives: NDT score, gradient and (Gauss-Newton, the svn default) Hessian of the source at the pose
  // p = [x, y, z, roll, pitch, yaw].  The reference takes the ORIGINAL points from its member input_ (set by align) and
  // the moved ones from `trans_cloud`; here the source is what setInputSource() / the last align() handed over and the
  // engine moves it itself -- trans_cloud is only checked for its size.  One launch (ndt_eval_derivatives, K = 1).
  template <class CloudPtr>
  void setInputSource(const CloudPtr& cloud) {
    if (!h_) return;
    status_ = cloud && !cloud->points.empty()
                  ? ndt_set_source(h_, &cloud->points[0].x, cloud->points.size(), sizeof(cloud->points[0]))
                  : ndt_set_source(h_, nullptr, 0, 12);
    n_source_ = status_ == NDT_OK && cloud ? cloud->points.size() : 0;
  }
  template <class Vec6, class Mat6, class Cloud>
  double computeParticleDerivatives(Vec6& score_gradient, Mat6& hessian, const Cloud& trans_cloud, const Vec6& p,
                                    bool compute_hessian = true) {
    double pose[6], words[NDT_EVAL_WORDS], score = 0.0, g[6], H[36];
    for (int i = 0; i < 6; ++i) { pose[i] = p[i]; score_gradient[i] = 0.0; }
    for (int r = 0; r < 6; ++r)
      for (int c = 0; c < 6; ++c) hessian(r, c) = 0.0;
    if (!h_) { status_ = NDT_ERR_NO_DEVICE; return 0.0; }
    if (trans_cloud.points.size() != n_source_) { status_ = NDT_ERR_INVALID_ARG; return 0.0; }
    status_ = ndt_eval_derivatives(h_, pose, nullptr, 1, compute_hessian ? 1 : 0, words);
    if (status_ != NDT_OK) return 0.0;
    ndt_unpack_eval(words, &score, g, H);
    for (int i = 0; i < 6; ++i) score_gradient[i] = g[i];
    if (compute_hessian)
      for (int r = 0; r < 6; ++r)
        for (int c = 0; c < 6; ++c) hessian(r, c) = H[6 * r + c];
    return score;
  }

  // ---- the reference's other public math hooks (ref: svn_ndt.h:208-276, svn_ndt_impl.hpp:214-500) ----
  // One point, one point-voxel pair, one pair of particles at a time, on the HOST, in the reference's own precision
  // (f32 products, f64 accumulation): the reference opens them to its tests, and so does the adapter -- the engine's
  // evaluation (k_derivatives; computeParticleDerivatives above) does not go through them.  The angle tables and the
  // Gaussian constants are the engine's (ndt_angle_tables, ndt_gauss_constants: what every launch receives), so a sum
  // of updateDerivatives over a cloud's pairs checks the kernel against the reference-shaped arithmetic
  // (tests/cpp/test_grid_queries.cpp).  Matrix arguments: anything with (row, col); vectors: anything with [i].
  void updateNdtConstants() { ndt_gauss_constants((double)prm_.resolution, prm_.outlier_ratio, &gauss_d1_, &gauss_d2_); }
  double gaussD1() const { return gauss_d1_; }
  double gaussD2() const { return gauss_d2_; }
  template <class Vec6>
  void computeAngleDerivatives(const Vec6& p, bool compute_hessian = true) {
    double pose[6];
    for (int i = 0; i < 6; ++i) pose[i] = p[i];
    ndt_angle_tables(pose, j_ang_, h_ang_);
    if (!compute_hessian) std::fill(h_ang_, h_ang_ + 45, 0.0f);
  }
  // Fills the ANGULAR entries of the 4 x 6 point Jacobian (the caller has set the translation block to identity, as in
  // the reference) and the 24 x 6 stack of second derivatives: block i (rows 4 i .. 4 i + 3), column j = d2 x' / dp_i dp_j.
  template <class Vec3, class PointGradient, class PointHessian>
  void computePointDerivatives(const Vec3& x, PointGradient& point_gradient, PointHessian& point_hessian, bool compute_hessian = true) const {
    const float xf[3] = {(float)x[0], (float)x[1], (float)x[2]};
    auto along = [&](const float* row) { return row[0] * xf[0] + row[1] * xf[1] + row[2] * xf[2]; };
    static const int where_j[8][2] = {{1, 3}, {2, 3}, {0, 4}, {1, 4}, {2, 4}, {0, 5}, {1, 5}, {2, 5}};   // (component, parameter)
    for (int k = 0; k < 8; ++k) point_gradient(where_j[k][0], where_j[k][1]) = along(j_ang_ + 3 * k);
    if (!compute_hessian) return;
    for (int r = 0; r < 24; ++r)
      for (int c = 0; c < 6; ++c) point_hessian(r, c) = 0.0f;
    // (parameter i, parameter j >= i, component): the table's rows in order; the roll-roll block has no x component
    static const int where_h[15][3] = {{3, 3, 1}, {3, 3, 2}, {3, 4, 1}, {3, 4, 2}, {3, 5, 1}, {3, 5, 2}, {4, 4, 0}, {4, 4, 1},
                                       {4, 4, 2}, {4, 5, 0}, {4, 5, 1}, {4, 5, 2}, {5, 5, 0}, {5, 5, 1}, {5, 5, 2}};
    for (int k = 0; k < 15; ++k) {
      const int i = where_h[k][0], j = where_h[k][1], comp = where_h[k][2];
      const float v = along(h_ang_ + 3 * k);
      point_hessian(4 * i + comp, j) = v;
      if (i != j) point_hessian(4 * j + comp, i) = v;
    }
  }
  // One point-voxel pair: adds its share to score_gradient / hessian and returns its score (Magnusson eq. 6.9, 6.12, 6.13).
  template <class Vec6, class Mat6, class PointGradient, class PointHessian, class Vec3, class Mat3>
  double updateDerivatives(Vec6& score_gradient, Mat6& hessian, const PointGradient& point_gradient4, const PointHessian& point_hessian,
                           const Vec3& x_trans, const Mat3& c_inv, bool compute_hessian = true, bool use_gauss_newton_hessian = true) const {
    const double d[3] = {x_trans[0], x_trans[1], x_trans[2]};
    double cd[3];
    for (int r = 0; r < 3; ++r) cd[r] = c_inv(r, 0) * d[0] + c_inv(r, 1) * d[1] + c_inv(r, 2) * d[2];
    double q = d[0] * cd[0] + d[1] * cd[1] + d[2] * cd[2];
    if (!std::isfinite(q) || q < -1e-9) return 0.0;
    if (q < 0.0) q = 0.0;
    const double arg = gauss_d2_ * q * 0.5;
    if (arg > 50.0) return 0.0;
    const double e = std::exp(-arg), score_inc = -gauss_d1_ * e, factor = gauss_d1_ * gauss_d2_ * e;
    if (!std::isfinite(factor) || std::fabs(factor) < 1e-15) return score_inc;
    const float df[3] = {(float)d[0], (float)d[1], (float)d[2]};
    float cj[3][6], row[6];   // C^-1 J (f32), (x - mu)^T C^-1 J
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 6; ++c)
        cj[r][c] = (float)c_inv(r, 0) * point_gradient4(0, c) + (float)c_inv(r, 1) * point_gradient4(1, c) + (float)c_inv(r, 2) * point_gradient4(2, c);
    bool fin = true;
    for (int c = 0; c < 6; ++c) {
      row[c] = df[0] * cj[0][c] + df[1] * cj[1][c] + df[2] * cj[2][c];
      fin = fin && std::isfinite(factor * (double)row[c]);
    }
    if (fin)
      for (int c = 0; c < 6; ++c) score_gradient[c] += factor * (double)row[c];
    if (!compute_hessian) return score_inc;
    double add[6][6];
    float dc[3];   // (x - mu)^T C^-1 (f32)
    for (int c = 0; c < 3; ++c) dc[c] = df[0] * (float)c_inv(0, c) + df[1] * (float)c_inv(1, c) + df[2] * (float)c_inv(2, c);
    fin = true;
    for (int i = 0; i < 6; ++i)
      for (int j = 0; j < 6; ++j) {
        const float jcj = point_gradient4(0, i) * cj[0][j] + point_gradient4(1, i) * cj[1][j] + point_gradient4(2, i) * cj[2][j];
        double t = (double)jcj;
        if (!use_gauss_newton_hessian) {
          const int a = i < j ? i : j, b = i < j ? j : i;   // (the stack is filled for both orders; the reference reads the upper one)
          const float third = dc[0] * point_hessian(4 * a + 0, b) + dc[1] * point_hessian(4 * a + 1, b) + dc[2] * point_hessian(4 * a + 2, b);
          t += -gauss_d2_ * ((double)row[i] * (double)row[j]) + (double)third;
        }
        add[i][j] = t * factor;
        fin = fin && std::isfinite(add[i][j]);
      }
    if (fin)
      for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) hessian(i, j) += add[i][j];
    return score_inc;
  }
  // k(l, k) = exp(-|Log(l^-1 k)|^2 / h) and its gradient with respect to l in l's tangent space ([rotation, translation],
  // gtsam's order), evaluated by the engine's own Stage-2 arithmetic (ndt_svn_rbf_kernel).  Pose3: anything with matrix();
  // h: setKernelBandwidth.  The gradient comes back as Vec6 (default: six doubles in a std::array).
  template <class Pose3>
  double rbf_kernel(const Pose3& pose_l, const Pose3& pose_k) const {
    double a[16], b[16], k = 0.0;
    pose_matrix(pose_l, a);
    pose_matrix(pose_k, b);
    ndt_svn_rbf_kernel(a, b, svn_.kernel_bandwidth, &k, nullptr);
    return k;
  }
  template <class Vec6 = std::array<double, 6>, class Pose3>
  Vec6 rbf_kernel_gradient(const Pose3& pose_l, const Pose3& pose_k) const {
    double a[16], b[16], k = 0.0, g[6];
    pose_matrix(pose_l, a);
    pose_matrix(pose_k, b);
    ndt_svn_rbf_kernel(a, b, svn_.kernel_bandwidth, &k, g);
    Vec6 out{};
    for (int i = 0; i < 6; ++i) out[i] = g[i];
    return out;
  }

  int lastStatus() const { return status_; }
  std::string lastError() const { return h_ ? ndt_last_error(h_) : "no engine (ndt_create failed: GPU required)"; }
  ndt_handle* handle() { return h_; }

 private:
  void push() { if (h_) status_ = ndt_set_params(h_, &prm_); updateNdtConstants(); }
  template <class Pose3>
  static void pose_matrix(const Pose3& p, double T[16]) {
    const auto M = p.matrix();
    for (int c = 0; c < 4; ++c)
      for (int r = 0; r < 4; ++r) T[4 * c + r] = M(r, c);
  }
  double gauss_d1_ = 0.0, gauss_d2_ = 0.0;
  float j_ang_[24] = {}, h_ang_[45] = {};
  size_t n_source_ = 0;
  ndt_params prm_{};
  ndt_svn_params svn_{};
  ndt_handle* h_ = nullptr;
  int status_ = NDT_OK;
  uint64_t seed_ = 1;
};

}  // namespace ndt_hip
