// Drives the adapter the way the reference's drivers do, through the reference's OWN names
// (`pclomp::...`, `svn_ndt::...`, `pcl::Registration::Ptr`) resolved by include/compat/, with the
// value types the drivers use (Eigen matrices, gtsam::Pose3).  Each block follows one call site:
//   [A] run/pipeline.cpp:464-481      engine creation, setters, upcast into RegisterCallback::registration
//   [B] run/pipeline.cpp:557-568      setInputTarget / setInputSource / align / getFinalTransformation / getResult
//   [C] run/pipeline.cpp:594-604      lidarCov = -(hessian + 1e-6 I)^-1, diagonal sqrt
//   [D] run/pipeline_ligo_tc.cpp:293,531   regularisation setters
//   [E] include/pipeline.hpp:163-222  extractNdtData(): leaves, Eigen-valued accessors, getLeafCenter
//   [F] run/pipeline_lo_svn.cpp:299-320,387-388   svn engine, align(cloud, gtsam::Pose3) -> SvnNdtResult
//   [G] extern/svn_ndt/test/test_svn_ndt.cpp:144-199   the PclOmp convergence test's calls and assertions
// Eigen / PCL / GTSAM here are the API mocks of tests/cpp/mock (test doubles, see its README):
// this file proves that the typed faces compile and run on the GPU, not that the real libraries
// accept them.  Exit code 0 = pass.  Run by tests/test_gpu_cpp_adapter.py.
#include <Eigen/Dense>
#include <gtsam/geometry/Pose3.h>
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
#include <pcl/registration/registration.h>

// what include/registercallback.hpp:7-17 includes
#include <pclomp/ndt_omp.h>
#include <pclomp/ndt_omp_impl.hpp>
#include <pclomp/voxel_grid_covariance_omp.h>
#include <pclomp/voxel_grid_covariance_omp_impl.hpp>
#include <svn_ndt.h>
#include <svn_ndt_impl.hpp>
#include <voxel_grid_covariance.h>
#include <voxel_grid_covariance_impl.hpp>

#include <cmath>
#include <cstdio>
#include <memory>
#include <random>
#include <string>
#include <vector>

static_assert(NDT_HIP_WITH_EIGEN && NDT_HIP_WITH_PCL && NDT_HIP_WITH_GTSAM, "all three faces must be on in this test");

using PointT = pcl::PointXYZI;
using Cloud = pcl::PointCloud<PointT>;

// the member of RegisterCallback the engine is stored in (ref: include/registercallback.hpp:35)
struct RegisterCallbackLike {
  pcl::Registration<pcl::PointXYZI, pcl::PointXYZI>::Ptr registration;
  int num_threads_ = 8;
  float ndt_resolution_ = 1.0f;
  float ndt_transform_epsilon_ = 1e-4f;
  std::string ndt_neighborhood_search_method_ = "DIRECT7";
  float regularization_scale_factor_ = 10.0f;
};

// the structs extractNdtData() fills (ref: include/pipeline.hpp:144-155)
struct NdtEllipsoid {
  Eigen::Vector3d mean;
  Eigen::Matrix3d evecs;
  Eigen::Vector3d evals;
  size_t point_count;
};
struct NdtVoxel {
  Eigen::Vector3d center;
  float resolution;
};

template <typename P, typename NDT_Type>
static size_t extract_like_pipeline_hpp(NDT_Type ndt, std::vector<NdtEllipsoid>& ellipsoids, std::vector<NdtVoxel>& voxels) {
  using TargetGrid = pclomp::VoxelGridCovariance<P>;                 // [E] :175
  const TargetGrid& target_cells = ndt->getTargetCells();            //     :178
  auto leaves = target_cells.getLeaves();                            //     :180
  size_t min_points = target_cells.getMinPointPerVoxel();            //     :181
  float resolution = ndt->getResolution();                           //     :184
  for (auto const& [index, leaf] : leaves) {                         //     :191
    if (leaf.getPointCount() >= (int)min_points) {
      ellipsoids.push_back(NdtEllipsoid{.mean = leaf.getMean(), .evecs = leaf.getEvecs(), .evals = leaf.getEvals(),
                                        .point_count = static_cast<size_t>(leaf.getPointCount())});
      voxels.push_back(NdtVoxel{.center = target_cells.getLeafCenter(index), .resolution = resolution});
    }
  }
  return leaves.size();
}

int main() {
  // ---- fixture: the two planes of test_svn_ndt.cpp:44-83, ground truth :104-106 ----
  const double cz = std::cos(0.2618), sz = std::sin(0.2618), cy = std::cos(0.0873), sy = std::sin(0.0873);
  Eigen::Matrix4d gt = Eigen::Matrix4d::Identity();
  gt(0, 0) = cz * cy; gt(0, 1) = -sz; gt(0, 2) = cz * sy; gt(0, 3) = 0.5;
  gt(1, 0) = sz * cy; gt(1, 1) = cz;  gt(1, 2) = sz * sy; gt(1, 3) = 0.0;
  gt(2, 0) = -sy;     gt(2, 1) = 0.0; gt(2, 2) = cy;      gt(2, 3) = 0.3;
  const double a = -0.03;
  Eigen::Matrix4d d = Eigen::Matrix4d::Identity();
  d(0, 0) = std::cos(a); d(0, 1) = -std::sin(a); d(1, 0) = std::sin(a); d(1, 1) = std::cos(a);
  d(0, 3) = -0.02; d(1, 3) = 0.01; d(2, 3) = -0.03;
  const Eigen::Matrix4d lidarFactorSourceTb2m = gt * d;  // the initial guess

  Cloud::Ptr pointsBody(new Cloud());
  Cloud::Ptr lidarFactorPointsTarget(new Cloud());
  std::mt19937 gen(1337);
  std::normal_distribution<double> noise(0.0, 0.02);
  for (int plane = 0; plane < 2; ++plane)
    for (double u = -10.0; u <= 10.0; u += 0.15)
      for (double v = -10.0; v <= 10.0; v += 0.15) {
        PointT p;
        p.x = (float)u; p.y = plane ? 0.0f : (float)v; p.z = plane ? (float)v : 0.0f;
        pointsBody->push_back(p);
        PointT q;
        q.x = (float)(gt(0, 0) * p.x + gt(0, 1) * p.y + gt(0, 2) * p.z + gt(0, 3) + noise(gen));
        q.y = (float)(gt(1, 0) * p.x + gt(1, 1) * p.y + gt(1, 2) * p.z + gt(1, 3) + noise(gen));
        q.z = (float)(gt(2, 0) * p.x + gt(2, 1) * p.y + gt(2, 2) * p.z + gt(2, 3) + noise(gen));
        lidarFactorPointsTarget->push_back(q);
      }

  bool ok = true;
  RegisterCallbackLike registerCallback;

  // ---- [A] run/pipeline.cpp:464-481 ----
  pclomp::NormalDistributionsTransform<pcl::PointXYZI, pcl::PointXYZI>::Ptr ndt_omp = nullptr;
  ndt_omp.reset(new pclomp::NormalDistributionsTransform<pcl::PointXYZI, pcl::PointXYZI>());
  if (ndt_omp->lastStatus() != NDT_OK) { std::printf("FAIL: engine: %s\n", ndt_omp->lastError().c_str()); return 2; }
  ndt_omp->setNumThreads(registerCallback.num_threads_);
  ndt_omp->setResolution(registerCallback.ndt_resolution_);
  ndt_omp->setTransformationEpsilon(registerCallback.ndt_transform_epsilon_);
  if (registerCallback.ndt_neighborhood_search_method_ == "DIRECT1") {
    ndt_omp->setNeighborhoodSearchMethod(pclomp::DIRECT1);
  } else if (registerCallback.ndt_neighborhood_search_method_ == "DIRECT7") {
    ndt_omp->setNeighborhoodSearchMethod(pclomp::DIRECT7);
  } else if (registerCallback.ndt_neighborhood_search_method_ == "KDTREE") {
    ndt_omp->setNeighborhoodSearchMethod(pclomp::KDTREE);
  }
  registerCallback.registration = ndt_omp;  // upcast to pcl::Registration
  // [G] the convergence test's extra setters
  ndt_omp->setMaximumIterations(50);
  ndt_omp->setStepSize(0.1);

  // ---- [B] run/pipeline.cpp:557-568 ----
  Cloud::Ptr lidarFactorPointsSource(new Cloud());
  registerCallback.registration->setInputTarget(lidarFactorPointsTarget);
  registerCallback.registration->setInputSource(pointsBody);
  registerCallback.registration->align(*lidarFactorPointsSource, lidarFactorSourceTb2m.cast<float>());
  Eigen::Matrix4d registerResult = registerCallback.registration->getFinalTransformation().cast<double>();
  auto ndt_result = ndt_omp->getResult();
  int ndt_iter = ndt_result.iteration_num;
  double terr = 0, tr = 0;
  for (int r = 0; r < 3; ++r) terr += (registerResult(r, 3) - gt(r, 3)) * (registerResult(r, 3) - gt(r, 3));
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) tr += registerResult(r, c) * gt(r, c);
  const double rerr = std::acos(std::fmin(1.0, std::fmax(-1.0, (tr - 1.0) / 2.0)));
  std::printf("[B] converged=%d iterations=%d trans_err=%.5f rot_err=%.5f kdtree_builds=%d\n",
              (int)registerCallback.registration->hasConverged(), ndt_iter, std::sqrt(terr), rerr,
              registerCallback.registration->kdtree_builds_);
  ok = ok && registerCallback.registration->hasConverged() && ndt_iter > 0 && ndt_iter < 50 && std::sqrt(terr) < 0.05 &&
       rerr < 0.035;                                    // [G] :185-198
  ok = ok && ndt_omp->getFinalNumIteration() == ndt_iter;
  ok = ok && registerCallback.registration->kdtree_builds_ == 0;  // the adapter spares align() the FLANN build

  // ---- [C] run/pipeline.cpp:594-604 ----
  const auto& hessian = ndt_result.hessian;
  Eigen::Matrix<double, 6, 6> regularized_hessian = hessian + (Eigen::Matrix<double, 6, 6>::Identity() * 1e-6);
  Eigen::Matrix<double, 6, 6> lidarCov = -regularized_hessian.inverse();
  Eigen::Matrix<double, 6, 1> lidarStdDev = lidarCov.diagonal().cwiseSqrt();
  Eigen::Matrix<double, 6, 6> viaAbi;
  const bool cov_ok = ndt_result.covarianceForGtsam(viaAbi, 1e-6, /*gtsam_order=*/false);
  double dcov = 0;
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j) dcov = std::fmax(dcov, std::fabs(viaAbi(i, j) - lidarCov(i, j)) / std::fabs(lidarCov(i, i)));
  std::printf("[C] std dev: %.2e %.2e %.2e m  %.2e %.2e %.2e rad; |cov - abi| = %.1e\n", lidarStdDev(0), lidarStdDev(1),
              lidarStdDev(2), lidarStdDev(3), lidarStdDev(4), lidarStdDev(5), dcov);
  ok = ok && cov_ok && dcov < 1e-9 && lidarStdDev(0) > 0 && lidarStdDev(5) > 0 && hessian(0, 0) < 0.0;

  // ---- [D] run/pipeline_ligo_tc.cpp:293,531-532 ----
  ndt_omp->setRegularizationScaleFactor(registerCallback.regularization_scale_factor_);
  ndt_omp->setRegularizationPose(lidarFactorSourceTb2m.cast<float>());
  ndt_omp->align(*lidarFactorPointsSource, lidarFactorSourceTb2m.cast<float>());
  ok = ok && ndt_omp->lastStatus() == NDT_OK;
  ndt_omp->unsetRegularizationPose();

  // ---- [E] include/pipeline.hpp:163-222 ----
  std::vector<NdtEllipsoid> ellipsoids;
  std::vector<NdtVoxel> voxels;
  const size_t n_leaves = extract_like_pipeline_hpp<PointT>(ndt_omp, ellipsoids, voxels);
  double worst = 0;
  for (size_t i = 0; i < ellipsoids.size(); ++i)
    for (int k = 0; k < 3; ++k) worst = std::fmax(worst, std::fabs(ellipsoids[i].mean(k) - voxels[i].center(k)));
  std::printf("[E] leaves=%zu exported=%zu max|mean - centre|=%.3f evals[0]=(%.2e %.2e %.2e)\n", n_leaves,
              ellipsoids.size(), worst, ellipsoids[0].evals(0), ellipsoids[0].evals(1), ellipsoids[0].evals(2));
  ok = ok && n_leaves > 100 && ellipsoids.size() == n_leaves && worst <= 0.5 + 1e-6 &&
       ellipsoids[0].evals(0) <= ellipsoids[0].evals(1) && ellipsoids[0].evals(1) <= ellipsoids[0].evals(2);

  // ---- [F] run/pipeline_lo_svn.cpp:299-320,387-388 ----
  std::unique_ptr<svn_ndt::SvnNormalDistributionsTransform<pcl::PointXYZI, pcl::PointXYZI>> svn_ndt_ptr = nullptr;
  svn_ndt_ptr = std::make_unique<svn_ndt::SvnNormalDistributionsTransform<pcl::PointXYZI, pcl::PointXYZI>>();
  svn_ndt_ptr->setResolution(1.0f);
  svn_ndt_ptr->setParticleCount(8);
  svn_ndt_ptr->setMaxIterations(100);
  svn_ndt_ptr->setKernelBandwidth(1.0);
  svn_ndt_ptr->setStepSize(1.0);
  svn_ndt_ptr->setEarlyStopThreshold(1e-4);
  svn_ndt_ptr->setOutlierRatio(0.55);
  svn_ndt_ptr->setNeighborhoodSearchMethod(svn_ndt::NeighborSearchMethod::DIRECT7);
  svn_ndt_ptr->setMinPointPerVoxel(3);
  svn_ndt_ptr->setParticleSeed(2);
  svn_ndt_ptr->setInputTarget(lidarFactorPointsTarget);
  const gtsam::Pose3 ins_pose(lidarFactorSourceTb2m);
  svn_ndt::SvnNdtResult result = svn_ndt_ptr->align(*pointsBody, ins_pose);
  const Eigen::Matrix4d F = result.final_pose.matrix();
  double serr = 0;
  for (int r = 0; r < 3; ++r) serr += (F(r, 3) - gt(r, 3)) * (F(r, 3) - gt(r, 3));
  std::printf("[F] svn converged=%d iterations=%d trans_err=%.5f cov(0,0)=%.3g\n", (int)result.converged,
              result.iterations, std::sqrt(serr), result.final_covariance(0, 0));
  ok = ok && result.iterations > 0 && std::sqrt(serr) < 0.05 && result.final_covariance(0, 0) > 0.0;

  // the particle kernel hooks (ref: svn_ndt.h:256-276): k(l, l) = 1 with a zero gradient; for a pure translation d the
  // logarithm is d itself: k = exp(-|d|^2 / h), gradient = k (-2 / h) [0, 0, 0, d] in gtsam's [rotation, translation] order
  {
    Eigen::Matrix4d A = Eigen::Matrix4d::Identity(), B = Eigen::Matrix4d::Identity();
    B(0, 3) = 0.3; B(1, 3) = -0.2; B(2, 3) = 0.1;
    const gtsam::Pose3 pa(A), pb(B);
    const double kaa = svn_ndt_ptr->rbf_kernel(pa, pa), kab = svn_ndt_ptr->rbf_kernel(pa, pb);
    const auto gaa = svn_ndt_ptr->rbf_kernel_gradient(pa, pa), gab = svn_ndt_ptr->rbf_kernel_gradient(pa, pb);
    const double want = std::exp(-(0.09 + 0.04 + 0.01) / 1.0);
    double gerr = std::fabs(gab[0]) + std::fabs(gab[1]) + std::fabs(gab[2]) + std::fabs(gab[3] + 2.0 * want * 0.3) +
                  std::fabs(gab[4] - 2.0 * want * 0.2) + std::fabs(gab[5] + 2.0 * want * 0.1);
    for (int i = 0; i < 6; ++i) gerr += std::fabs(gaa[i]);
    std::printf("[F] rbf_kernel: k(a, a)=%.3f k(a, b)=%.12f (want %.12f) gradient err %.1e; gauss d1=%.6f d2=%.6f\n", kaa, kab, want, gerr,
                svn_ndt_ptr->gaussD1(), svn_ndt_ptr->gaussD2());
    ok = ok && kaa == 1.0 && std::fabs(kab - want) < 1e-15 && gerr < 1e-14 && svn_ndt_ptr->gaussD1() < 0.0 && svn_ndt_ptr->gaussD2() > 0.0;
  }

  std::printf(ok ? "PASS\n" : "FAIL\n");
  return ok ? 0 : 1;
}
