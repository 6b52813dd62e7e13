// ndt_newton.h -- host side of align(): Newton iterations with a More-Thuente
// line search on the 6-vector pose, driving an abstract derivative evaluator
// (the HIP kernels in production).  Internal header.
#pragma once

#include <functional>
#include <vector>

#include "../../include/ndt_hip.h"

namespace ndt {

// One global evaluation (already summed over shards / GPUs).
struct Eval {
  double score = 0;
  double g[6] = {0, 0, 0, 0, 0, 0};
  double H[36];  // row-major, symmetric
  double nvtl_sum = 0;
  double n_with = 0;
  double n_pairs = 0;
};

// pose6 -> f32 4x4 (column-major), R = Rx*Ry*Rz built in f32
void pose_to_matrix(const double p[6], float T[16]);
// inverse, angles as Eigen's eulerAngles(0,1,2) returns them
void matrix_to_pose(const float T[16], double p[6]);
// Gauss constants d1, d2 (ref: extern/svn_ndt/include/svn_ndt_impl.hpp:80-131)
void gauss_constants(double resolution, double outlier_ratio, double* d1, double* d2);
// angle tables (ref: svn_ndt_impl.hpp:255-334)
void angle_tables(const double p[6], float jang[24], float hang[45]);
// expands NDT_EVAL_WORDS packed words into an Eval
void unpack_eval(const double* w, Eval* e);
// adds the ridge / regularisation terms and applies the non-finite guards
void finish_eval(const ndt_params& prm, const float* reg_pose, const double p[6], bool need_h, Eval* e);

// fn(pose6, T, need_hessian, out) -> 0 on success
using EvalFn = std::function<int(const double*, const float*, bool, Eval*)>;

// pclomp::NdtResult's per-iteration arrays [RECALLED: tier4 ndt_omp, absent submodule]: entry 0 is the initial guess
// (transform only: its score arrays hold the first evaluation), one entry per Newton iteration after it
struct IterHistory {
  std::vector<float> transforms;            // 16 floats per entry, column-major
  std::vector<double> transform_probability, nvtl;
  void clear() { transforms.clear(); transform_probability.clear(); nvtl.clear(); }
  void push(const float T[16], double tp, double nv) {
    transforms.insert(transforms.end(), T, T + 16);
    transform_probability.push_back(tp);
    nvtl.push_back(nv);
  }
  size_t size() const { return transform_probability.size(); }
};

// hessian_in_trials: ask for the Hessian in every line-search trial instead of one extra
// evaluation at the accepted step (same numbers -- the extra evaluation is at the pose of
// the last trial -- one launch fewer per Newton iteration that needed trials).
int newton_align(const ndt_params& prm, int64_t n_source_total, const float guess[16],
                 const EvalFn& fn, ndt_result* out, bool hessian_in_trials = false, IterHistory* history = nullptr);

// cov = -(H + eps I)^-1, optionally with the [rotation, translation] block order of a GTSAM
// Pose3 noise model (ref: run/pipeline.cpp:594-596, src/registercallback.cpp:170-186).
// Returns false when H + eps I is singular or not finite.
bool result_covariance(const double H[36], double eps, bool gtsam_order, double cov[36]);

}  // namespace ndt
