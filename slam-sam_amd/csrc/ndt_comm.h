// ndt_comm.h -- cross-GPU reduction of one derivative evaluation (internal).
//
// The source cloud is sharded across the GPUs of a node (one process per GPU),
// the voxel table is replicated; the only exchange is a sum of the
// NDT_EVAL_WORDS-double partial (256 B) once per evaluation.
#pragma once

#include <hip/hip_runtime.h>

#include <string>

#include "../../include/ndt_hip.h"
#include "ndt_device.h"

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
  // ranks of the live reducer as the transport itself reports them (RCCL: ncclCommCount of the
  // communicator), 1 without a reducer, < 0 on failure
  int rank_count() const;
  int init_rccl(const void* id128, int rank, int nranks, std::string* err);
  int init_shm(const char* name, int rank, int nranks, std::string* err);
  int init_hook(ndt_allreduce_fn fn, void* ctx, int rank, int nranks);
  // NDT_REDUCE_P2P (XchgInfo in ndt_device.h): p2p_handle allocates this rank's exchange area on the
  // current device and exports its IPC handle (NDT_P2P_HANDLE_BYTES); init_p2p maps every rank's area
  // (handles: nranks x NDT_P2P_HANDLE_BYTES in rank order; the own entry is not opened)
  int p2p_handle(void* out64, std::string* err);
  int init_p2p(const void* handles, int rank, int nranks, std::string* err);
  // the XchgInfo the derivative kernel reads (device memory), and the tag of the NEXT global evaluation
  const XchgInfo* p2p_info() const { return static_cast<const XchgInfo*>(xinfo_dev_); }
  uint64_t p2p_round() const { return xround_; }
  void p2p_set_round(uint64_t r) { xround_ = r; }
  // {exchanges made inside a kernel, their summed duration, the longest (10 ns ticks), exchanges a peer was late for}
  // since init_p2p; `reset`: back to zero afterwards
  int p2p_stats(unsigned long long out[4], bool reset, std::string* err);
  // COLLECTIVE slot-integrity pass over the exchange areas (every rank calls it with the same `rounds`): `rounds` lock-step
  // rounds of patterned {tag, value} slots, the value a function of (tag, writer, word), written to every rank's area and
  // read back by every rank exactly as the derivative kernel's final sum does it (16-byte system-scope stores and loads, two
  // generations) -- the only direct test of the assumption that such a slot is seen entirely old or entirely new.
  // out: {rounds completed, slots whose tag was new and whose value was not (torn), rounds a peer did not show up in time
  // (the pass stops at the first), longest wait for a round in 10 ns ticks}
  int p2p_selftest(int rounds, unsigned long long out[4], std::string* err);
  // host side of the exchange: the rows of round `round` of the own area, waited for (120 s) and added in
  // rank order -- the continuation of a kernel that reported EV_FAIL = 3 (its own row is published)
  int p2p_finish_on_host(uint64_t round, double* words, int n, std::string* err);
  void destroy();

  // true when the partial must be produced in DEVICE memory (RCCL reduces there)
  bool wants_device_buffer() const { return mode_ == NDT_REDUCE_RCCL; }

  // device-side all-reduce, in place, enqueued on `s` (RCCL mode only)
  int allreduce_device(double* d_words, int n, hipStream_t s, std::string* err);
  // host-side all-reduce of words that are already on the host, in place
  int allreduce_host(double* words, int n, std::string* err);
  // K evaluations of NDT_EVAL_WORDS words each (a batched launch's results), in place.  NDT_REDUCE_P2P exchanges up to
  // XCHG_BATCH_MAX of them in ONE round (one publish kernel, one strided read-back per poll) instead of K rounds of
  // blocking copies; the other transports go evaluation by evaluation.
  int allreduce_host_batch(double* words, int K, std::string* err);

 private:
  int mode_ = NDT_REDUCE_NONE;
  int rank_ = 0, nranks_ = 1;
  void* nccl_comm_ = nullptr;
  // shared-memory segment
  void* shm_ = nullptr;
  size_t shm_bytes_ = 0;
  std::string shm_name_;
  uint64_t shm_round_ = 0;
  // p2p
  void* xarea_ = nullptr;                 // own exchange area (fine-grained device memory)
  void* xpeer_[XCHG_MAX_RANKS] = {};      // every rank's area as mapped here; [rank_] == xarea_
  void* xinfo_dev_ = nullptr;
  uint64_t xround_ = 0;                   // global evaluations exchanged so far
  uint64_t xbround_ = 0;                  // batched rounds exchanged so far (their own region of the area, their own tags)
  void* xstats_ = nullptr;                // four counters (XchgInfo::stats), device memory
  uint64_t xtests_ = 0;                   // integrity passes run so far (their tags never repeat)
  double* xstage_ = nullptr;              // pinned, device-mapped: the batch's values on their way to the publish kernel
  unsigned long long* xback_ = nullptr;   // pinned: the rows of a batch round as read back
  int p2p_publish_from_host(uint64_t round, const double* words, int n, std::string* err);
  // hook
  ndt_allreduce_fn hook_ = nullptr;
  void* hook_ctx_ = nullptr;
};

}  // namespace ndt
