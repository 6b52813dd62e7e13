// What the drivers pay per align() THROUGH pcl::Registration (ref: run/pipeline.cpp:557-561, align is called on
// RegisterCallback::registration, a pcl::Registration::Ptr): the host work of pcl::Registration::align around
// computeTransformation -- identity indices, output.resize, a per-point copy of the 32-byte source points, the
// data[3] = 1 pass -- reproduced in the API mock (tests/cpp/mock/pcl/registration/registration.h) and timed here
// on a source of the headline size, next to the engine's own align.  Prints one JSON object; bench.py quotes it
// under host_cloud.  Usage: bench_registration_prework [n_source] [n_target] [aligns].
#include <Eigen/Dense>
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
#include <pcl/registration/registration.h>
#include <pclomp/ndt_omp.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>

using PointT = pcl::PointXYZI;
using Cloud = pcl::PointCloud<PointT>;

int main(int argc, char** argv) {
  const size_t n_src = argc > 1 ? (size_t)atoll(argv[1]) : 200000, n_tgt = argc > 2 ? (size_t)atoll(argv[2]) : 1000000;
  const int reps = argc > 3 ? atoi(argv[3]) : 30;
  // a room: floor + two walls, noisy; geometry is irrelevant to the pre-work (it only depends on n_src)
  std::mt19937 gen(7);
  std::uniform_real_distribution<float> U(-20.0f, 20.0f);
  std::normal_distribution<float> N(0.0f, 0.02f);
  auto make = [&](size_t n, float dx) {
    Cloud::Ptr c(new Cloud());
    c->points.reserve(n);
    for (size_t i = 0; i < n; ++i) {
      PointT p;
      const int w = (int)(i % 3);
      const float a = U(gen), b = U(gen);
      if (w == 0) { p.x = a; p.y = b; p.z = N(gen); }
      else if (w == 1) { p.x = a; p.y = 20.0f + N(gen); p.z = 0.2f * (b + 20.0f); }
      else { p.x = -20.0f + N(gen); p.y = a; p.z = 0.2f * (b + 20.0f); }
      p.x += dx;
      c->push_back(p);
    }
    c->width = (unsigned)n;
    return c;
  };
  Cloud::Ptr target = make(n_tgt, 0.0f), source = make(n_src, 0.1f);
  pclomp::NormalDistributionsTransform<PointT, PointT>::Ptr ndt(new pclomp::NormalDistributionsTransform<PointT, PointT>());
  if (ndt->lastStatus() != NDT_OK) { std::printf("{\"error\": \"%s\"}\n", ndt->lastError().c_str()); return 2; }
  ndt->setResolution(0.5f);
  ndt->setTransformationEpsilon(1e-4);
  ndt->setStepSize(0.1);
  ndt->setMaximumIterations(35);
  ndt->setNeighborhoodSearchMethod(pclomp::DIRECT7);
  pcl::Registration<PointT, PointT>::Ptr registration = ndt;   // RegisterCallback::registration
  registration->setInputTarget(target);
  registration->setInputSource(source);
  Cloud out;
  const Eigen::Matrix4f guess = Eigen::Matrix4f::Identity();
  for (int i = 0; i < 3; ++i) registration->align(out, guess);
  registration->prework_ns_ = 0; registration->prework_calls_ = 0;
  const auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < reps; ++i) registration->align(out, guess);
  const double ms_align_through_pcl = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / reps;
  const double ms_prework = 1e-6 * (double)registration->prework_ns_ / (double)registration->prework_calls_;
  // the same align without pcl::Registration in between (computeTransformation is public, as in pclomp:
  // ref extern/svn_ndt/test/test_svn_ndt.cpp:171)
  const auto t1 = std::chrono::steady_clock::now();
  for (int i = 0; i < reps; ++i) ndt->computeTransformation(out, guess);
  const double ms_align_direct = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count() / reps;
  std::printf("{\"n_source\": %zu, \"n_target\": %zu, \"aligns\": %d, \"ms_align_through_pcl_registration\": %.4f, "
              "\"ms_pcl_registration_prework\": %.4f, \"ms_align_direct\": %.4f, \"iterations\": %d, \"converged\": %d}\n",
              n_src, n_tgt, reps, ms_align_through_pcl, ms_prework, ms_align_direct, ndt->getResult().iteration_num,
              (int)registration->hasConverged());
  return 0;
}
