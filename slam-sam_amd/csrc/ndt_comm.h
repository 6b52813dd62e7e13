// ndt_comm.h -- cross-GPU reduction of one derivative evaluation (internal).
//
// The source cloud is sharded across the GPUs of a node (one process per GPU),
// the voxel table is replicated; the only exchange is a sum of the
// NDT_EVAL_WORDS-double partial (256 B) once per evaluation.
#pragma once

#include <hip/hip_runtime.h>

#include <string>

#include "../../include/ndt_hip.h"

namespace ndt {

class Reducer {
 public:
  Reducer() = default;
  ~Reducer() { destroy(); }

  int mode() const { return mode_; }
  int rank() const { return rank_; }
  int nranks() const { return nranks_; }

  static int unique_id(void* out128);
  // ncclGetVersion() and the shared object that serves the nccl* symbols of this library
  static int library_info(char* path_buf, size_t cap);
  int init_rccl(const void* id128, int rank, int nranks, std::string* err);
  int init_shm(const char* name, int rank, int nranks, std::string* err);
  int init_hook(ndt_allreduce_fn fn, void* ctx, int rank, int nranks);
  void destroy();

  // true when the partial must be produced in DEVICE memory (RCCL reduces there)
  bool wants_device_buffer() const { return mode_ == NDT_REDUCE_RCCL; }

  // device-side all-reduce, in place, enqueued on `s` (RCCL mode only)
  int allreduce_device(double* d_words, int n, hipStream_t s, std::string* err);
  // host-side all-reduce of words that are already on the host, in place
  int allreduce_host(double* words, int n, std::string* err);

 private:
  int mode_ = NDT_REDUCE_NONE;
  int rank_ = 0, nranks_ = 1;
  void* nccl_comm_ = nullptr;
  // shared-memory segment
  void* shm_ = nullptr;
  size_t shm_bytes_ = 0;
  std::string shm_name_;
  uint64_t shm_round_ = 0;
  // hook
  ndt_allreduce_fn hook_ = nullptr;
  void* hook_ctx_ = nullptr;
};

}  // namespace ndt
