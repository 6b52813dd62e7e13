// The reference's voxel-grid query surface and svn_ndt's public math hook on the adapter, through the reference's own
// names (include/compat), against the CPU oracle (test infrastructure): neighbour sets of getNeighborhoodAtPoint7 / 1 and
// radiusSearch, getLeaf(point), getCentroids, nearestKSearch, getCovEigValueInflationRatio
// (ref: extern/svn_ndt/include/voxel_grid_covariance.h:194-200,280-381) and computeParticleDerivatives
// (ref: extern/svn_ndt/include/svn_ndt.h:186-206) and, summed over a cloud's pairs, the per-pair hooks
// computeAngleDerivatives / computePointDerivatives / updateDerivatives (:208-254).  Exit code 0 = pass.  Needs an MI355X; run by tests/test_gpu_cpp_adapter.py.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <random>
#include <set>
#include <vector>

#include <pclomp/ndt_omp.h>
#include <svn_ndt.h>
#include <voxel_grid_covariance.h>

#include "../../oracle/ndt_oracle.h"

using PointT = ndt_hip::PointXYZI;
using Cloud = ndt_hip::PointCloud<PointT>;

#define CHECK(c) do { if (!(c)) { std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

struct V6 { double v[6]; double& operator[](int i) { return v[i]; } const double& operator[](int i) const { return v[i]; } };

int main() {
  // a room-like target: floor, two walls, noise; 1 m voxels
  std::mt19937 gen(7);
  std::normal_distribution<float> noise(0.0f, 0.03f);
  std::uniform_real_distribution<float> uni(-12.0f, 12.0f);
  auto tgt = std::make_shared<Cloud>();
  for (int i = 0; i < 60000; ++i) {
    const float u = uni(gen), v = uni(gen);
    const int k = i % 3;
    PointT p{};
    p.x = k == 1 ? 9.3f + noise(gen) : u;
    p.y = k == 2 ? -7.6f + noise(gen) : (k == 1 ? v : u * 0.5f + v * 0.5f);
    // (every surface has two independent in-plane coordinates: a voxel whose points lie on a LINE has a covariance of rank
    // one plus rounding, and whether its smallest eigenvalue comes out as +1e-18 or -1e-18 -- kept or rejected, ref
    // voxel_grid_covariance_impl.hpp:318-327 -- is the eigen-solver's business, not this test's)
    p.z = k == 0 ? 0.2f + noise(gen) : (k == 1 ? 1.5f + 0.12f * u : 1.5f + 0.12f * v);
    tgt->points.push_back(p);
  }
  for (int i = 0; i < 4; ++i) {   // a voxel with too few points to be a leaf
    PointT p{};
    p.x = -10.5f + 0.1f * i; p.y = 10.5f; p.z = 4.5f + 0.05f * i;
    tgt->points.push_back(p);
  }
  pclomp::NormalDistributionsTransform<PointT, PointT> ndt;
  if (ndt.lastStatus() != NDT_OK) { std::printf("FAIL: engine: %s\n", ndt.lastError().c_str()); return 2; }
  ndt.setResolution(1.0f);
  ndt.setInputTarget(tgt);
  const svn_ndt::VoxelGridCovariance<PointT>& grid = ndt.getTargetCells();
  CHECK(ndt.lastStatus() == NDT_OK && grid.getLeaves().size() > 300);
  CHECK(grid.getCovEigValueInflationRatio() == 0.01 && grid.getMinPointPerVoxel() == 6);

  oracle_params prm;
  oracle_default_params(&prm);
  prm.resolution = 1.0f;
  oracle_grid* og = oracle_grid_build(&tgt->points[0].x, tgt->points.size(), sizeof(PointT), &prm);
  oracle_grid_info gi;
  oracle_grid_get_info(og, &gi);
  // getLeaves(): every occupied voxel, like the reference's map; the oracle exports the VALID ones (>= 6 points, invertible)
  std::vector<const ndt_hip::TargetGrid::Entry*> valid;
  for (const auto& e : grid.getLeaves())
    if (e.second.getPointCount() >= grid.getMinPointPerVoxel()) valid.push_back(&e);
  std::printf("occupied voxels %zu (oracle %lld), valid leaves %zu (oracle %lld)\n", grid.getLeaves().size(), (long long)gi.n_cells_hit,
              valid.size(), (long long)gi.n_leaves);
  CHECK((size_t)gi.n_leaves == valid.size() && grid.getLeaves().size() <= (size_t)gi.n_cells_hit && valid.size() < (size_t)gi.n_cells_hit);
  {
    PointT sparse{};
    sparse.x = -10.4f; sparse.y = 10.5f; sparse.z = 4.55f;
    CHECK(grid.getLeaf(sparse) == nullptr);   // (four points: occupied, not a leaf)
  }
  std::vector<int64_t> cell(gi.n_leaves);
  std::vector<int32_t> cnt(gi.n_leaves);
  std::vector<double> mean(3 * gi.n_leaves), cov(9 * gi.n_leaves), icov(9 * gi.n_leaves), evecs(9 * gi.n_leaves), evals(3 * gi.n_leaves);
  oracle_grid_export(og, cell.data(), cnt.data(), mean.data(), cov.data(), icov.data(), evecs.data(), evals.data());
  for (size_t i = 0; i < cell.size(); ++i) {   // same leaves, same order; only valid ones through getLeaf(index)
    CHECK((int64_t)valid[i]->first == cell[i] && grid.getLeaf(valid[i]->first) == &valid[i]->second);
  }
  for (const auto& e : grid.getLeaves())
    if (e.second.getPointCount() < grid.getMinPointPerVoxel()) CHECK(grid.getLeaf(e.first) == nullptr);
  CHECK(grid.getCentroids().size() == cell.size());
  for (size_t i = 0; i < cell.size(); i += 17) CHECK(grid.getCentroids()[i].x == (float)mean[3 * i]);

  auto ranks_of = [&](const std::vector<const ndt_hip::TargetGrid::Leaf*>& ls) {
    std::multiset<int64_t> s;
    for (const auto* l : ls) s.insert((int64_t)l->d.index);
    return s;
  };
  auto oracle_set = [&](const PointT& q, int method) {
    int64_t r[27];
    const float p[3] = {q.x, q.y, q.z};
    const int n = oracle_grid_neighbors(og, p, method, r);
    std::multiset<int64_t> s;
    for (int k = 0; k < n; ++k) s.insert(cell[(size_t)r[k]]);
    return s;
  };
  std::vector<const ndt_hip::TargetGrid::Leaf*> nb;
  std::vector<float> d2;
  long n7 = 0, n1 = 0, nr = 0;
  std::uniform_real_distribution<float> q(-14.0f, 14.0f);   // some queries outside the grid
  for (int i = 0; i < 4000; ++i) {
    PointT p{};
    if (i % 4 == 0) {   // on a voxel face, where the f32 re-classification of the offset point matters
      p.x = std::floor(q(gen)); p.y = q(gen); p.z = 0.5f * q(gen) * 0.2f;
    } else {
      p = tgt->points[(size_t)(gen() % tgt->points.size())];
      p.x += 0.4f * noise(gen) * 10.0f; p.y += 0.4f * noise(gen) * 10.0f; p.z += 0.2f * noise(gen) * 10.0f;
    }
    CHECK(grid.getNeighborhoodAtPoint7(p, nb) == (int)nb.size());
    CHECK(ranks_of(nb) == oracle_set(p, ORACLE_DIRECT7));
    n7 += (long)nb.size();
    CHECK(grid.getNeighborhoodAtPoint1(p, nb) <= 1);
    CHECK(ranks_of(nb) == oracle_set(p, ORACLE_DIRECT1));
    n1 += (long)nb.size();
    const ndt_hip::TargetGrid::Leaf* direct = grid.getLeaf(p);
    CHECK((direct != nullptr) == (nb.size() == 1) && (!direct || direct == nb[0]));
    const float pv[3] = {p.x, p.y, p.z};
    struct Vec { const float* a; float operator[](int k) const { return a[k]; } const float* data() const { return a; } } v3{pv};
    CHECK(grid.getLeaf(v3) == direct);   // the Eigen::Vector3f face
    CHECK(grid.radiusSearch(p, 1.0, nb, d2) == (int)nb.size() && d2.size() == nb.size());
    CHECK(ranks_of(nb) == oracle_set(p, ORACLE_KDTREE));
    CHECK(std::is_sorted(d2.begin(), d2.end()));
    for (float d : d2) CHECK(d < 1.0f);
    nr += (long)nb.size();
    if (i % 50 == 0) {
      std::vector<const ndt_hip::TargetGrid::Leaf*> kn;
      std::vector<float> kd;
      CHECK(grid.nearestKSearch(p, 5, kn, kd) == 5 && std::is_sorted(kd.begin(), kd.end()));
      if (!nb.empty()) CHECK(kn[0] == nb[0] && kd[0] == d2[0]);   // the nearest centroid heads both lists
      CHECK(grid.radiusSearch(p, 1.0, nb, d2, 2) <= 2);
    }
  }
  CHECK(n7 > 4000 && n1 > 1000 && nr > 3000);
  std::printf("neighbour sets agree with the oracle on 4000 queries: DIRECT7 %ld, DIRECT1 %ld, radius %ld leaves\n", n7, n1, nr);

  // ---- svn_ndt's computeParticleDerivatives ----
  auto src = std::make_shared<Cloud>();
  for (size_t i = 0; i < tgt->points.size(); i += 5) src->points.push_back(tgt->points[i]);
  svn_ndt::SvnNormalDistributionsTransform<PointT, PointT> svn;
  CHECK(svn.lastStatus() == NDT_OK);
  svn.setResolution(1.0f);
  svn.setInputTarget(tgt);
  svn.setInputSource(src);
  V6 p{{0.05, -0.03, 0.02, 0.004, -0.003, 0.01}}, g{};
  ndt_hip::Matrix6d H;
  Cloud moved = *src;   // (the engine moves the source itself; only the size is looked at)
  const double score = svn.computeParticleDerivatives(g, H, moved, p, true);
  CHECK(svn.lastStatus() == NDT_OK);
  prm.hessian_mode = ORACLE_HESSIAN_GAUSS_NEWTON;
  prm.pair_mode = ORACLE_PAIR_SVN_F64;
  prm.num_threads = 4;
  float T[16];
  oracle_pose_to_matrix(p.v, T);
  oracle_derivs od;
  oracle_derivatives(og, &src->points[0].x, src->points.size(), sizeof(PointT), T, p.v, &prm, 1, &od);
  double gmax = 0, hmax = 0, gerr = 0, herr = 0;
  for (int i = 0; i < 6; ++i) { gmax = std::max(gmax, std::fabs(od.gradient[i])); gerr = std::max(gerr, std::fabs(g[i] - od.gradient[i])); }
  for (int r = 0; r < 6; ++r)
    for (int c = 0; c < 6; ++c) { hmax = std::max(hmax, std::fabs(od.hessian[6 * r + c])); herr = std::max(herr, std::fabs(H(r, c) - od.hessian[6 * r + c])); }
  std::printf("computeParticleDerivatives: score %.9f (oracle %.9f), gradient err %.2e of %.2e, Hessian err %.2e of %.2e\n", score, od.score,
              gerr, gmax, herr, hmax);
  CHECK(std::fabs(score - od.score) <= 1e-9 * std::fabs(od.score) && od.n_pairs > 10000);
  CHECK(gerr <= 1e-9 * gmax && herr <= 1e-9 * hmax);
  Cloud wrong;
  svn.computeParticleDerivatives(g, H, wrong, p, true);
  CHECK(svn.lastStatus() == NDT_ERR_INVALID_ARG);

  // ---- the per-pair hooks (ref: svn_ndt.h:208-254): computeAngleDerivatives once, then for every source point
  // computePointDerivatives and, for every neighbour of the moved point, updateDerivatives -- the loop of the reference's
  // computeParticleDerivatives (svn_ndt_impl.hpp:518-668) written with the adapter's names.  Its sums must agree with the
  // engine's one-launch evaluation to the f32 products the hooks use (measured 5e-9 .. 3e-8 of the largest entry; bound 1e-6),
  // Gauss-Newton and full.
  for (int full = 0; full < 2; ++full) {
    svn.setUseGaussNewtonHessian(full == 0);
    V6 g_dev{};
    ndt_hip::Matrix6d H_dev;
    const double s_dev = svn.computeParticleDerivatives(g_dev, H_dev, moved, p, true);
    CHECK(svn.lastStatus() == NDT_OK);
    svn.computeAngleDerivatives(p, true);
    V6 g_host{};
    ndt_hip::Matrix6d H_host;
    for (int r = 0; r < 6; ++r)
      for (int c = 0; c < 6; ++c) H_host(r, c) = 0.0;
    double s_host = 0.0;
    long pairs = 0;
    ndt_hip::Mat<float, 4, 6> pg;
    ndt_hip::Mat<float, 24, 6> ph;
    for (const PointT& q : src->points) {
      for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 6; ++c) pg(r, c) = (r == c && r < 3) ? 1.0f : 0.0f;
      const double x[3] = {q.x, q.y, q.z};
      svn.computePointDerivatives(x, pg, ph, true);
      PointT m{};   // the moved point, f32 in the reference's association (transformPointCloud)
      m.x = T[0] * q.x + (T[4] * q.y + (T[8] * q.z + T[12]));
      m.y = T[1] * q.x + (T[5] * q.y + (T[9] * q.z + T[13]));
      m.z = T[2] * q.x + (T[6] * q.y + (T[10] * q.z + T[14]));
      std::vector<const ndt_hip::TargetGrid::Leaf*> nb;
      grid.getNeighborhoodAtPoint7(m, nb);
      for (const auto* leaf : nb) {
        const ndt_hip::Vector3d mu = leaf->getMean();
        const double d[3] = {(double)m.x - mu[0], (double)m.y - mu[1], (double)m.z - mu[2]};
        const ndt_hip::Matrix3d ci = leaf->getInverseCov();
        s_host += svn.updateDerivatives(g_host, H_host, pg, ph, d, ci, true, full == 0);
        ++pairs;
      }
    }
    for (int i = 0; i < 6; ++i) H_host(i, i) += 1e-6;   // the ridge of computeParticleDerivatives (ref :650-653)
    double gm = 0, hm = 0, ge = 0, he = 0;
    for (int i = 0; i < 6; ++i) { gm = std::max(gm, std::fabs(g_dev[i])); ge = std::max(ge, std::fabs(g_host[i] - g_dev[i])); }
    for (int r = 0; r < 6; ++r)
      for (int c = 0; c < 6; ++c) { hm = std::max(hm, std::fabs(H_dev(r, c))); he = std::max(he, std::fabs(H_host(r, c) - H_dev(r, c))); }
    std::printf("per-pair hooks, %s Hessian: %ld pairs, score %.9f (engine %.9f), gradient err %.2e of %.2e, Hessian err %.2e of %.2e\n",
                full ? "full" : "Gauss-Newton", pairs, s_host, s_dev, ge, gm, he, hm);
    CHECK(pairs == od.n_pairs);
    CHECK(std::fabs(s_host - s_dev) <= 1e-9 * std::fabs(s_dev) && ge <= 1e-6 * gm && he <= 1e-6 * hm);
  }
  oracle_grid_free(og);
  std::printf("PASS\n");
  return 0;
}
