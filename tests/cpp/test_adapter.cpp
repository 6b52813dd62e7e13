// The reference's ConvergenceComparison.PclOmp test
// (ref: extern/svn_ndt/test/test_svn_ndt.cpp:138-199) written against the C++ adapter's
// PCL-free face.  Exit code 0 = pass.  Needs an MI355X; run by tests/test_gpu_cpp_adapter.py.
#include <array>
#include <cmath>
#include <cstdio>
#include <random>

#include "ndt_hip/ndt_hip.hpp"

using PointT = ndt_hip::PointXYZ;
using Cloud = ndt_hip::PointCloud<PointT>;

static void matmul4(const double* A, const double* B, double* C) {  // column-major
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r) {
      double s = 0;
      for (int k = 0; k < 4; ++k) s += A[4 * k + r] * B[4 * c + k];
      C[4 * c + r] = s;
    }
}

int main() {
  // ground truth Rz(0.2618) Ry(0.0873), t = (0.5, 0, 0.3); guess = gt with a small error
  const double cz = std::cos(0.2618), sz = std::sin(0.2618), cy = std::cos(0.0873), sy = std::sin(0.0873);
  double gt[16] = {cz * cy, sz * cy, -sy, 0, -sz, cz, 0, 0, cz * sy, sz * sy, cy, 0, 0.5, 0.0, 0.3, 1};
  const double a = -0.03;  // small yaw + translation error
  double d[16] = {std::cos(a), std::sin(a), 0, 0, -std::sin(a), std::cos(a), 0, 0, 0, 0, 1, 0, -0.02, 0.01, -0.03, 1};
  double guess_d[16];
  matmul4(gt, d, guess_d);

  auto src = std::make_shared<Cloud>();
  auto tgt = std::make_shared<Cloud>();
  std::mt19937 gen(1337);
  std::normal_distribution<double> noise(0.0, 0.02);
  for (int plane = 0; plane < 2; ++plane)
    for (double u = -10.0; u <= 10.0; u += 0.15)
      for (double v = -10.0; v <= 10.0; v += 0.15) {
        PointT p{(float)u, plane ? 0.0f : (float)v, plane ? (float)v : 0.0f, 1.0f};
        src->points.push_back(p);
        double q[3];
        for (int r = 0; r < 3; ++r) q[r] = gt[r] * p.x + gt[4 + r] * p.y + gt[8 + r] * p.z + gt[12 + r] + noise(gen);
        tgt->points.push_back(PointT{(float)q[0], (float)q[1], (float)q[2], 1.0f});
      }

  ndt_hip::NormalDistributionsTransform<PointT, PointT> ndt;
  if (ndt.lastStatus() != NDT_OK) { std::printf("FAIL: engine: %s\n", ndt.lastError().c_str()); return 2; }
  ndt.setResolution(1.0f);
  ndt.setNeighborhoodSearchMethod(ndt_hip::DIRECT7);
  ndt.setMaximumIterations(50);
  ndt.setTransformationEpsilon(1e-4);
  ndt.setStepSize(0.1);
  ndt.setNumThreads(20);
  ndt.setInputTarget(tgt);
  ndt.setInputSource(src);
  ndt_hip::Matrix4f guess;
  for (int i = 0; i < 16; ++i) guess[i] = (float)guess_d[i];
  Cloud out;
  ndt.setFillOutputCloud(true);
  ndt.computeTransformation(out, guess);
  const bool converged = ndt.hasConverged();
  const int iters = ndt.getFinalNumIteration();
  ndt_hip::Matrix4f T = ndt.getFinalTransformation();
  double terr = 0, tr = 0;
  for (int r = 0; r < 3; ++r) terr += (T[12 + r] - gt[12 + r]) * (T[12 + r] - gt[12 + r]);
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) tr += T[4 * c + r] * gt[4 * c + r];  // trace(R^T Rgt)
  const double rerr = std::acos(std::fmin(1.0, std::fmax(-1.0, (tr - 1.0) / 2.0)));
  auto res = ndt.getResult();
  const auto& cells = ndt.getTargetCells();
  std::printf("converged=%d iterations=%d trans_err=%.5f rot_err=%.5f leaves=%zu out=%zu H00=%.3g nvtl=%.3f\n",
              (int)converged, iters, std::sqrt(terr), rerr, cells.getLeaves().size(), out.size(), res.hessian[0],
              res.nearest_voxel_transformation_likelihood);
  bool ok = converged && iters < 50 && std::sqrt(terr) < 0.05 && rerr < 0.035;
  ok = ok && res.iteration_num == iters && res.hessian[0] < 0.0 && !cells.getLeaves().empty();
  ok = ok && out.size() == src->size();
  // failure path: aligning an engine without clouds returns the guess, not converged, no throw
  ndt_hip::NormalDistributionsTransform<PointT, PointT> empty;
  Cloud o2;
  empty.computeTransformation(o2, guess);
  ok = ok && !empty.hasConverged() && empty.lastStatus() == NDT_ERR_NO_TARGET &&
       empty.getFinalTransformation() == guess;
  // keyframe archive: target = keyframe 1 (the target cloud, identity pose), source = keyframe 2
  // -> the same registration, with both clouds resident on the device; plus the covariance helper
  {
    ndt_hip::NormalDistributionsTransform<PointT, PointT> kf;
    kf.setResolution(1.0f);
    kf.setMaximumIterations(50);
    kf.setTransformationEpsilon(1e-4);
    kf.setStepSize(0.1);
    kf.putKeyframe(1, *tgt);
    kf.putKeyframe(2, *src);
    const double eye[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    kf.setInputTargetFromKeyframes({1}, eye);
    kf.setInputSourceFromKeyframe(2);
    Cloud o3;
    kf.computeTransformation(o3, guess);
    ndt_hip::Matrix4f T2 = kf.getFinalTransformation();
    double d2 = 0;
    for (int i = 0; i < 16; ++i) d2 = std::fmax(d2, std::fabs((double)T2[i] - (double)T[i]));
    ndt_hip::Matrix6d cov{};
    const bool cov_ok = kf.getResult().covarianceForGtsam(cov);
    std::printf("keyframes: status=%d converged=%d max|T - T_host|=%.2e cov_ok=%d cov[rot x]=%.3g\n", kf.lastStatus(),
                (int)kf.hasConverged(), d2, (int)cov_ok, cov[0]);
    ok = ok && kf.lastStatus() == NDT_OK && kf.hasConverged() && d2 < 1e-5 && kf.keyframeCount() == 2 && cov_ok && cov[0] > 0.0;
  }
  // scoring-only calls + the O(log V) leaf lookup
  {
    const double tp = ndt.calculateTransformationProbability(*src, T);
    const double nv = ndt.calculateNearestVoxelTransformationLikelihood(*src, T);
    const auto& leaves = cells.getLeaves();
    const auto& mid = leaves[leaves.size() / 2];
    const ndt_hip::Vector3d c = cells.getLeafCenter(mid.first);
    const ndt_hip::Vector3d mu = mid.second.getMean();
    double off = 0;
    for (int a = 0; a < 3; ++a) off = std::fmax(off, std::fabs(c[a] - mu[a]));
    std::printf("score-only: tp=%.4f nvtl=%.4f (align: %.4f %.4f); leaf %zu centre-mean offset %.3f\n", tp, nv,
                ndt.getTransformationProbability(), ndt.getNearestVoxelTransformationLikelihood(), mid.first, off);
    ok = ok && std::fabs(tp - ndt.getTransformationProbability()) < 1e-9 * std::fabs(tp) + 1e-12 &&
         std::fabs(nv - ndt.getNearestVoxelTransformationLikelihood()) < 1e-9 && off <= 0.5 + 1e-6 &&
         cells.getLeaf(mid.first) == &mid.second && cells.getLeaf((size_t)-1) == nullptr;
  }
  // multi-grid face [RECALLED tier4 names]: the target split into two tiles with string ids
  {
    ndt_hip::MultiGridNormalDistributionsTransform<PointT, PointT> mg;
    mg.setResolution(1.0f);
    mg.setMaximumIterations(50);
    mg.setTransformationEpsilon(1e-4);
    mg.setStepSize(0.1);
    auto left = std::make_shared<Cloud>(), right = std::make_shared<Cloud>();
    for (const auto& p : tgt->points) {
      if (p.x < 1.0f) left->points.push_back(p);
      if (p.x > -1.0f) right->points.push_back(p);
    }
    mg.addTarget(left, "tile_left");
    mg.addTarget(right, "tile_right");
    mg.createVoxelKdtree();
    const int st0 = mg.lastStatus();
    mg.setInputSource(src);
    Cloud o4;
    mg.computeTransformation(o4, guess);
    ndt_hip::Matrix4f T4 = mg.getFinalTransformation();
    double e4 = 0;
    for (int r = 0; r < 3; ++r) e4 += (T4[12 + r] - gt[12 + r]) * (T4[12 + r] - gt[12 + r]);
    const size_t union_leaves = mg.getTargetCells().getLeaves().size();
    mg.removeTarget("tile_left");
    Cloud o5;
    mg.computeTransformation(o5, guess);   // the union is gone until it is re-created
    const int st1 = mg.lastStatus();
    std::printf("multigrid: create=%d converged_err=%.5f union_leaves=%zu ids=%zu after_remove_status=%d\n", st0, std::sqrt(e4),
                union_leaves, mg.getCurrentMapIDs().size(), st1);
    ok = ok && st0 == NDT_OK && std::sqrt(e4) < 0.05 && union_leaves > cells.getLeaves().size() &&
         mg.getCurrentMapIDs().size() == 1 && st1 == NDT_ERR_NO_TARGET;
  }
  {  // pcl::VoxelGrid on the device (ref: run/pipeline_ins_map_distribution.cpp:324-340), handshake + wait() of the hand-off
    ndt_hip::NormalDistributionsTransform<PointT, PointT> vg;
    Cloud ds;
    vg.voxelDownsample(*tgt, 0.5f, ds);
    bool inside = !ds.points.empty() && ds.points.size() < tgt->points.size();
    for (size_t i = 0; i < ds.points.size() && inside; ++i) inside = std::isfinite(ds.points[i].x) && std::isfinite(ds.points[i].z);
    vg.setResolution(1.0f);
    vg.setInputTarget(tgt);
    vg.setInputTarget(tgt);            // steady state: asynchronous
    const int w = vg.wait();
    std::printf("voxelDownsample: %zu -> %zu points, status %d; wait() = %d\n", tgt->points.size(), ds.points.size(), vg.lastStatus(), w);
    ok = ok && inside && w == NDT_OK;
  }
  // svn_ndt-shaped adapter: K = 8 particles, Gauss-Newton Hessian, one launch per iteration
  ndt_hip::SvnNormalDistributionsTransform<PointT, PointT> svn;
  svn.setResolution(1.0f);
  svn.setMinPointPerVoxel(3);
  svn.setNeighborhoodSearchMethod(ndt_hip::DIRECT7);
  svn.setParticleCount(8);
  svn.setMaxIterations(100);
  svn.setKernelBandwidth(1.0);
  svn.setEarlyStopThreshold(1e-4);
  svn.setStepSize(1.0);
  svn.setParticleSeed(2);
  svn.setInputTarget(tgt);
  ndt_hip::SvnNdtResult sr = svn.align(*src, guess_d);
  double serr = 0;
  for (int r = 0; r < 3; ++r) serr += (sr.final_pose[12 + r] - gt[12 + r]) * (sr.final_pose[12 + r] - gt[12 + r]);
  std::printf("svn: converged=%d iterations=%d trans_err=%.5f cov00=%.3g\n", (int)sr.converged, sr.iterations,
              std::sqrt(serr), sr.final_covariance[0]);
  ok = ok && sr.iterations > 0 && std::sqrt(serr) < 0.1 && sr.final_covariance[0] > 0.0;
  std::printf(ok ? "PASS\n" : "FAIL\n");
  return ok ? 0 : 1;
}
