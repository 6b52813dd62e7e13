// ndt_evaluate.hip -- derivative evaluations (ordinary, pre-launched, batched), align, scoring (see ndt_engine.h).
#include "ndt_engine.h"

namespace ndt {
namespace engine {

// Kernel timing (ndt_enable_kernel_timing): the two events are attached to the derivative kernel's dispatch, so their
// difference is the kernel's own duration -- the figure rocprofv3 reports.  ndt_tuning::timing_bracket = 1 records them
// around the launch call instead, as rounds 1-2 did (adds the dispatch, ~2.4 us: the tuning scripts' older numbers).
bool timing_brackets_launch() { return tuning().timing_bracket != 0; }

// ndt_tuning::prelaunch_probe = 0: the automatic stream placement never probes the other placement (A/B aid)
bool auto_probe_enabled() { return tuning().prelaunch_probe != 0; }

// (re)writes the packed copy of the record table; `wait`: the caller is about to launch on another stream
int pack_records(ndt_handle* h, bool wait) {
  if (h->n_slots <= 0) return NDT_OK;
  HIP_TRY(h, h->prec.ensure((size_t)h->n_slots));
  launch_pack_records(h->rec.p, h->prec.p, (size_t)h->n_slots, h->stream);
  HIP_TRY(h, hipGetLastError());
  if (wait) HIP_TRY(h, hipStreamSynchronize(h->stream));
  h->prec_valid = true;
  return NDT_OK;
}

// The record table an evaluation reads: the 48-byte packed copy when that format is selected and the neighbourhood is
// DIRECT7 / DIRECT1 (a multi-grid union chains its leaves through VoxelRecord::pad; the 27-cell neighbourhoods gained
// nothing from it), the 80-byte f64 records otherwise.
int records_for_eval(ndt_handle* h, EvalConsts* ec, const VoxelRecord** rec) {
  ec->packed = 0;
  *rec = h->rec.p;
  if (h->record_format != NDT_RECORDS_PACKED48 || h->multi_active || h->n_slots <= 0 ||
      (h->prm.search_method != NDT_DIRECT7 && h->prm.search_method != NDT_DIRECT1))
    return NDT_OK;  // (an empty table: slot 0 of the f64 one is what absent neighbours read)
  if (!h->prec_valid) {
    int rc = pack_records(h, true);  // (the format was selected after the build, or the table came from another path)
    if (rc) return rc;
  }
  ec->packed = 1;
  *rec = reinterpret_cast<const VoxelRecord*>(h->prec.p);
  return NDT_OK;
}

void fill_pose_consts(const double p[6], const float T[16], PoseConsts* pc) {
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) pc->R[3 * i + j] = T[4 * j + i];
    pc->t[i] = T[12 + i];
  }
  angle_tables(p, pc->jang, pc->hang);
}

// Per-process random bits for the LDS tags of k_derivatives (ndt_derivs.hip, "finishing waves"): two processes that
// share a device count their launch sequence numbers from the same start.
unsigned long long process_item_salt() {
  static const unsigned long long salt = [] {
    std::random_device rd;
    unsigned long long v = ((unsigned long long)rd() << 32) ^ (unsigned long long)rd();
    v ^= (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count() * 0x9E3779B97F4A7C15ull;
    return v;
  }();
  return salt;
}

EvalConsts make_eval_consts(const ndt_handle* h, bool need_h) {
  EvalConsts ec{};
  gauss_constants((double)h->prm.resolution, h->prm.outlier_ratio, &ec.d1, &ec.d2);
  ec.direct7 = h->prm.search_method == NDT_DIRECT7 ? 1 : 0;
  ec.kdtree = h->prm.search_method == NDT_KDTREE ? 1 : 0;
  ec.direct26 = h->prm.search_method == NDT_DIRECT26 ? 1 : 0;
  ec.score_only = 0;
  ec.kd_radius2 = (float)((double)h->prm.resolution * (double)h->prm.resolution);
  ec.need_hessian = need_h ? 1 : 0;
  ec.gauss_newton = h->prm.hessian_mode == NDT_HESSIAN_GAUSS_NEWTON ? 1 : 0;
  ec.multigrid = h->multi_active ? 1 : 0;
  ec.mbox_tagged = h->mbox_tagged ? 1 : 0;
  ec.mbox_preload = h->mbox_preload ? 1 : 0;
  ec.compute_units = h->n_cus;   // block shapes and the XCD count are those of THIS handle's device (CPX partitions: 32)
  // (a rank of a multi-rank job may share its device with the other ranks -- the one-device rehearsals do --, and blocks
  // that insist on a compute unit of their own would keep a peer's blocks, which this rank's kernel may be waiting for, out)
  ec.own_units = h->red.mode() == NDT_REDUCE_NONE ? 1 : 0;
  ec.item_salt = process_item_salt();
  return ec;
}

int ensure_counters(ndt_handle* h, size_t k) {
  if (k <= h->counters_zeroed) return NDT_OK;
  HIP_TRY(h, h->counters.ensure(k));
  HIP_TRY(h, hipMemsetAsync(h->counters.p, 0, h->counters.cap * sizeof(unsigned int), h->stream));
  h->counters_zeroed = h->counters.cap;
  return NDT_OK;
}

int ready_for_eval(ndt_handle* h) {
  int rc = settle(h);
  if (rc) return rc;
  if (!h->have_grid || h->n_valid <= 0) return fail(h, NDT_ERR_NO_TARGET, "no target voxel grid (setInputTarget first)");
  if (h->n_src == 0 && h->red.mode() == NDT_REDUCE_NONE) return fail(h, NDT_ERR_NO_SOURCE, "no source cloud (setInputSource first)");
  return NDT_OK;
}

// pinned result slots for K poses (32 tagged 16-byte slots each), zeroed when (re)allocated
int ensure_flag_slots(ndt_handle* h, size_t K) {
  if (K <= h->flag_slots && h->flag.h) return NDT_OK;
  HIP_TRY(h, h->flag.ensure(K * 2 * EV_WORDS));
  std::memset(h->flag.h, 0, K * 2 * EV_WORDS * sizeof(unsigned long long));
  h->flag_slots = K;
  return NDT_OK;
}

// The source in block order of the target grid under T, when the engine's parameters ask for it
// (NDT_SOURCE_ORDER_*).  Done once per (source, target): the copy stays a valid permutation of the
// source whatever the later poses are.
int maybe_sort_source(ndt_handle* h, const float T[16]) {
  if (h->src_sorted || h->n_src == 0 || !h->have_grid) return NDT_OK;
  const int mode = h->prm.source_order;
  if (mode == NDT_SOURCE_ORDER_KEEP) return NDT_OK;
  if (mode == NDT_SOURCE_ORDER_AUTO &&
      ((size_t)h->n_valid * sizeof(VoxelRecord) <= (size_t)6 << 20 || h->n_src < 32768))
    return NDT_OK;
  const size_t n = h->n_src;
  HIP_TRY(h, h->ox.ensure(n));
  HIP_TRY(h, h->oy.ensure(n));
  HIP_TRY(h, h->oz.ensure(n));
  HIP_TRY(h, h->skeys.ensure(n));
  HIP_TRY(h, h->skeys2.ensure(n));
  HIP_TRY(h, h->svals.ensure(n));
  HIP_TRY(h, h->svals2.ensure(n));
  HIP_TRY(h, h->ssort_tmp.ensure(sort_temp_bytes(n)));
  HIP_TRY(h, h->splan.ensure(1));
  PoseConsts pc{};
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) pc.R[3 * i + j] = T[4 * j + i];
    pc.t[i] = T[12 + i];
  }
  HIP_TRY(h, sort_source_by_blocks(h->vx, h->vy, h->vz, n, h->geom, pc, h->splan.p, h->ssort_tmp.p, h->skeys.p,
                                   h->skeys2.p, h->svals.p, h->svals2.p, h->ox.p, h->oy.p, h->oz.p, h->stream));
  h->src_sorted = true;
  return NDT_OK;
}

// would maybe_sort_source() sort this source for a grid of n_valid leaves?
bool source_sort_wanted(const ndt_handle* h, int n_valid) {
  if (h->src_sorted || h->n_src == 0) return false;
  const int mode = h->prm.source_order;
  if (mode == NDT_SOURCE_ORDER_KEEP) return false;
  if (mode == NDT_SOURCE_ORDER_AUTO && ((size_t)n_valid * sizeof(VoxelRecord) <= (size_t)6 << 20 || h->n_src < 32768)) return false;
  return true;
}

// An align whose target's build is still in flight (asynchronous hand-off, keyframe assembly, ndt_set_target_device_deferred)
// enqueues its first evaluation BEHIND that build instead of first waiting for the verdict and then paying a launch: the
// kernel takes the grid geometry from the build's device-side BuildGeom (launch_derivatives, d_geom) and leaves at once
// after a refused build.  The launch call and the dispatch run under the build (the first evaluation of an align cost 26 us
// against 16 for the later ones, tools/first_eval_cost.py).  Only where nothing else depends on the verdict: results
// polled from pinned slots, f64 records, no cross-rank exchange inside the kernel, a source that needs no re-ordering
// for a grid of the size of the previous one (checked again once the verdict is in: a mismatch discards the launch).
// ndt_tuning::speculate_first = 0: off.
bool first_eval_behind_build(const ndt_handle* h) {
  return tuning().speculate_first != 0 && h->spec_enabled && h->build_pending && h->prev_n_valid > 0 && h->n_src > 0 && !h->timing &&
         h->prm.wait_mode == NDT_WAIT_SPIN && !h->red.wants_device_buffer() && h->red.mode() != NDT_REDUCE_P2P &&
         h->record_format != NDT_RECORDS_PACKED48 && !source_sort_wanted(h, h->prev_n_valid);
}

// Launch sequence numbers tag every partial / result slot the derivative kernel writes; they
// must never repeat within the process (a freed partials buffer of one handle can become
// another's), hence one counter for all handles, starting at 1 (zeroed memory never matches).
std::atomic<unsigned long long> g_launch_seq{1};

// (re)allocates the partial rows; fresh memory is zeroed so that no slot carries a stale tag
int ensure_partials(ndt_handle* h, size_t words) {
  if (words <= h->partials.cap) return NDT_OK;
  HIP_TRY(h, h->partials.ensure(words));
  HIP_TRY(h, hipMemsetAsync(h->partials.p, 0, h->partials.cap * sizeof(double), h->stream));
  return NDT_OK;
}

// Wait for the derivative kernel's 32 result slots {seq, value} in pinned host memory (each
// slot is one 16-byte device store).  Polling them sees the result ~4 us sooner than
// hipStreamSynchronize (measured: 7.7 vs 11.5 us launch + completion round trip on MI355X);
// falls back to the stream if they never arrive.
bool slots_complete(const volatile unsigned long long* slots, unsigned long long seq) {
  for (int v = EV_WORDS - 1; v >= 0; --v)
    if (__atomic_load_n(slots + 2 * v, __ATOMIC_ACQUIRE) != seq) return false;
  return true;
}

// How long the host polls before it hands the wait to the runtime: every wait INSIDE the kernel is
// bounded by MBOX_TIMEOUT_TICKS (20 ms: a pre-launched kernel waiting for its pose), and the launch
// itself runs for tens of microseconds, so slots that have not appeared after 3 x that bound are not
// "late" -- the kernel is queued behind foreign work, or it is gone.  hipStreamSynchronize is always
// safe (it loses nothing): afterwards the slots are either there or the launch has failed.
constexpr auto kHostSpinLimit = std::chrono::microseconds(3 * (MBOX_TIMEOUT_TICKS / 100));

void quit_prelaunched(ndt_handle* h);

int wait_slots(ndt_handle* h, unsigned long long seq, int K, int first) {
  const volatile unsigned long long* f = h->flag.h + (size_t)first * 2 * EV_WORDS;
  const auto t0 = std::chrono::steady_clock::now();
  unsigned int spins = 0;
  auto all_complete = [&] {
    for (int k = K - 1; k >= 0; --k)
      if (!slots_complete(f + (size_t)k * 2 * EV_WORDS, seq)) return false;
    return true;
  };
  while (!all_complete()) {
    if ((++spins & 0x3FFF) == 0 && std::chrono::steady_clock::now() - t0 > kHostSpinLimit) {
      // the kernel enqueued for the NEXT evaluation is told to leave first: left waiting for its pose on the other
      // stream, it would hold this synchronisation for its own 20 ms and count as a time-out of its own
      quit_prelaunched(h);
      HIP_TRY(h, hipStreamSynchronize(h->stream));
      HIP_TRY(h, hipStreamSynchronize(h->stream2));
      if (!all_complete()) {
        h->counters_zeroed = 0;  // the ticket words may be stale: re-zero them before the next launch
        return fail(h, NDT_ERR_HIP, "derivative kernel finished without publishing its result");
      }
      break;
    }
  }
  return NDT_OK;
}

// ---- pre-launched evaluation -------------------------------------------------------------------
// The mailbox: fine-grained device memory the host can write by pointer (large BAR).  Absent
// (no large BAR, allocation refused, NDT_PRELAUNCH=0 in the environment) -> plain launches.
bool ensure_mailbox(ndt_handle* h) {
  if (h->mbox_tried) return h->mbox != nullptr;
  h->mbox_tried = true;
  const char* e = getenv("NDT_PRELAUNCH");
  if (e && atoi(e) == 0) return false;
  int largebar = 0;
  if (hipDeviceGetAttribute(&largebar, hipDeviceAttributeIsLargeBar, h->device) != hipSuccess || largebar != 1) return false;
  void* p = nullptr;
  if (hipExtMallocWithFlags(&p, 4096, hipDeviceMallocFinegrained) != hipSuccess || !p) {
    (void)hipGetLastError();
    return false;
  }
  if (hipMemset(p, 0, 4096) != hipSuccess) { (void)hipFree(p); return false; }
  h->mbox = static_cast<PoseMailbox*>(p);
  return true;
}

// The pose goes through write-combined BAR memory.  Tagged form (default): 82 granules of 8 bytes
// {launch tag, word}, ONE aligned 64-bit volatile store each (never split by the compiler, the CPU or
// a partially flushed write-combining buffer), one fence -- the kernel needs no second look after the
// tag.  Plain form (NDT_MBOX_TAGGED=0): pose first, fence, sequence number last, fence.
static inline void mbox_store_granule(PoseMailbox* m, int k, unsigned int tag, unsigned int word) {
  *reinterpret_cast<volatile unsigned long long*>(&m->gran[k][0]) = (unsigned long long)tag | ((unsigned long long)word << 32);
}

void publish_pose(ndt_handle* h, unsigned long long seq, const PoseConsts& pc) {
  static_assert(sizeof(PoseConsts) == 81 * sizeof(float), "PoseConsts is 81 packed floats");
  if (h->mbox_tagged) {
    unsigned int w[81];
    std::memcpy(w, &pc, sizeof(PoseConsts));
    const unsigned int tag = mbox_tag32(seq);
    for (int k = 0; k < 81; ++k) mbox_store_granule(h->mbox, k, tag, w[k]);
    mbox_store_granule(h->mbox, MBOX_GRANULES - 1, tag, 0u);
    _mm_sfence();
    return;
  }
  std::memcpy(const_cast<unsigned int*>(h->mbox->words), &pc, sizeof(PoseConsts));
  _mm_sfence();
  *reinterpret_cast<volatile unsigned long long*>(&h->mbox->seq) = seq;
  _mm_sfence();
}

// tells a waiting pre-launched kernel to leave (stream order does the rest)
void quit_prelaunched(ndt_handle* h) {
  if (h->pre_seq == 0) return;
  if (h->mbox_tagged) mbox_store_granule(h->mbox, MBOX_GRANULES - 1, mbox_tag32(h->pre_seq), MBOX_CTRL_QUIT);
  else *reinterpret_cast<volatile unsigned long long*>(&h->mbox->seq) = h->pre_seq | MBOX_QUIT;
  _mm_sfence();
  h->pre_seq = 0;
  h->n_prelaunch_quit++;
}

// one global evaluation at (p, T): local kernel + cross-rank sum
int evaluate(ndt_handle* h, const double p[6], const float T[16], bool need_h, Eval* out, bool score_only,
             bool safe_retry) {
  hipStream_t s = h->stream;
  bool speculate = false;
  if (h->spec_first) {   // ndt_align left the pending build's verdict to this call
    h->spec_first = false;
    speculate = first_eval_behind_build(h) && !score_only && !safe_retry && h->pre_seq == 0;
    if (!speculate) {
      int rc = ready_for_eval(h);
      if (rc) { h->spec_build_failed = true; return rc; }
      rc = maybe_sort_source(h, T);
      if (rc) return rc;
    } else {
      int rc = settle_source(h);   // the engine's streams ordered behind the source's transfer
      if (rc) return rc;
    }
  }
  PoseConsts pc;
  fill_pose_consts(p, T, &pc);
  EvalConsts ec = make_eval_consts(h, need_h);
  ec.score_only = score_only ? 1 : 0;
  ec.safe_sum = safe_retry ? 1 : 0;
#ifdef NDT_TEST_SEAMS
  {  // test seam (libndt_hip_seams.so only): one block of the N-th evaluation launch withholds its partial row
    static const int mute_at = [] { const char* e = getenv("NDT_DEBUG_MUTE_ROW_AT"); return e ? atoi(e) : -1; }();
    if (mute_at >= 0 && !safe_retry && h->tm.n_eval_launches == mute_at) ec.mute_row = 3;
  }
#endif
  const VoxelRecord* records = nullptr;
  {
    int rc = records_for_eval(h, &ec, &records);
    if (rc) return rc;
    rc = ensure_partials(h, derivs_partials_words(h->n_src, 1, h->n_cus));
    if (rc) return rc;
  }
  HIP_TRY(h, h->result.ensure(EV_WORDS));
  {
    int rc = ensure_flag_slots(h, 2);  // two result buffers, used in turn (below)
    if (rc) return rc;
  }
  const bool dev_out = h->red.wants_device_buffer();
  if (dev_out) HIP_TRY(h, h->dres.ensure(EV_WORDS));
  {
    int rc = ensure_counters(h, (size_t)derivs_counters_per_pose());
    if (rc) return rc;
  }
  double* d_out = dev_out ? h->dres.p : h->result.d;
  const bool spin = !dev_out && !h->timing && h->prm.wait_mode == NDT_WAIT_SPIN;
  const float* px = h->src_sorted ? h->ox.p : h->vx;
  const float* py = h->src_sorted ? h->oy.p : h->vy;
  const float* pz = h->src_sorted ? h->oz.p : h->vz;
  const bool prelaunch = spin && !score_only && !safe_retry && h->prelaunch_armed && !h->prelaunch_suspended &&
                         h->prm.prelaunch != NDT_PRELAUNCH_OFF && ensure_mailbox(h);
  unsigned long long seq = 0;
  bool via_mailbox = false;
  // NDT_REDUCE_P2P: the kernel's final sum exchanges the evaluation with the other ranks itself, under
  // the tag "number of this global evaluation" (identical on every rank: all run the same host loop on
  // the same sums).  A pre-launched kernel got its tag when it was enqueued; one that is told to leave
  // has consumed none.
  const bool p2p = h->red.mode() == NDT_REDUCE_P2P;
  const XchgInfo* xinfo = p2p ? h->red.p2p_info() : nullptr;
  unsigned long long xround = p2p ? h->red.p2p_round() + 1 : 0;
  // Consecutive single-pose launches write their results to ALTERNATING host buffers: a pre-launched
  // kernel runs ahead of the host, and one that gives up waiting for its pose (this thread frozen for
  // 20 ms -- a cgroup-throttled or oversubscribed host does that) writes its notice while the result
  // of its predecessor may still be unread.  With one shared buffer that notice replaced the unread
  // result and the host waited for tags that were gone ("finished without publishing", found by the
  // soak under host contention, tests/gpu_mbox_stress.py).
  int buf = 0;
  if (h->pre_seq != 0) {
    if (prelaunch && h->pre_need_h == need_h) {  // the kernel for this evaluation is already waiting on the device
      seq = h->pre_seq;
      buf = h->pre_buf;
      xround = h->pre_round;
      h->cur_on2 = h->pre_on2;
      h->pre_seq = 0;
#ifdef NDT_TEST_SEAMS
      {  // test seam (libndt_hip_seams.so only): hold the pose back so that the waiting kernel gives up
        static const int delay_ms = [] { const char* e = getenv("NDT_DEBUG_PUBLISH_DELAY_MS"); return e ? atoi(e) : 0; }();
        if (delay_ms > 0 && h->n_prelaunch_used == 3) std::this_thread::sleep_for(std::chrono::milliseconds(delay_ms));
      }
#endif
      publish_pose(h, seq, pc);
      via_mailbox = true;
      h->n_prelaunch_used++;
    } else {
      quit_prelaunched(h);
    }
  }
#ifdef NDT_TEST_SEAMS
  {  // test seam (libndt_hip_seams.so only): rank 1 of a peer-write job is late for one evaluation, so that the
     // other ranks' kernels give up waiting for its row (EV_FAIL = 3) and their hosts finish the exchange
    static const int late_ms = [] { const char* e = getenv("NDT_DEBUG_P2P_LATE_MS"); return e ? atoi(e) : 0; }();
    if (late_ms > 0 && p2p && h->red.rank() == 1 && h->tm.n_eval_launches == 4)
      std::this_thread::sleep_for(std::chrono::milliseconds(late_ms));
  }
#endif
  if (!via_mailbox) {
    seq = g_launch_seq.fetch_add(1, std::memory_order_relaxed);
    buf = (h->flag_toggle ^= 1);
    h->cur_on2 = 0;
    const bool bracket = h->timing && timing_brackets_launch();
    if (bracket) HIP_TRY(h, hipEventRecord(h->ev0, s));
    launch_derivatives(px, py, pz, h->n_src, h->geom, h->cell2leaf.p, records, h->cent.p, pc, nullptr, 1, ec, h->partials.p,
                       h->counters.p, d_out, s, spin ? h->flag.d + (size_t)buf * 2 * EV_WORDS : nullptr, seq, nullptr,
                       xinfo, xround, nullptr, nullptr, h->timing && !bracket ? h->ev0 : nullptr,
                       h->timing && !bracket ? h->ev1 : nullptr, speculate ? h->gd.p : nullptr);
    HIP_TRY(h, hipGetLastError());
    if (bracket) HIP_TRY(h, hipEventRecord(h->ev1, s));
  }
  if (speculate) {
    // the launch is on the stream behind the build; NOW the verdict (the host had nothing else to do meanwhile)
    const int rs = settle(h);
    const bool keep = rs == NDT_OK && h->have_grid && h->n_valid > 0 && h->brun.attempt == 0 && !source_sort_wanted(h, h->n_valid);
    if (!keep) {
      // refused or repeated build (the kernel left at once, or evaluated a grid that has been rebuilt since), no valid
      // voxel, or a grid for which the source is to be re-ordered: the launch is drained and forgotten
      ++h->n_spec_discarded;
      HIP_TRY(h, hipStreamSynchronize(s));
      h->counters_zeroed = 0;
      if (rs) { h->spec_build_failed = true; return rs; }
      int rc = ready_for_eval(h);
      if (rc) { h->spec_build_failed = true; return rc; }
      rc = maybe_sort_source(h, T);
      if (rc) return rc;
      return evaluate(h, p, T, need_h, out, score_only);
    }
    ++h->n_spec_used;
  }
  if (p2p) h->red.p2p_set_round(xround);  // this evaluation's tag is spent (a fallback below gives it back)
  if (prelaunch) {
    // the next evaluation's kernel goes onto the stream now, behind the one in flight; it will
    // start when that one has finished and wait for its pose (or for the order to leave)
    h->pre_seq = g_launch_seq.fetch_add(1, std::memory_order_relaxed);
    h->pre_need_h = need_h;
    h->pre_buf = (h->flag_toggle ^= 1);
    h->pre_round = xround + 1;
    // Which stream: behind the evaluation in flight (same stream: starts when that launch has ENDED), or on the
    // other stream, where its blocks take compute units as the blocks of the launch in flight leave -- resident
    // and polling by the time the host has the next pose (the end-of-launch barrier, the dispatch and the cold
    // start of a kernel are 2-3 us of every evaluation otherwise).  Only when the launch in flight needs no
    // more compute units itself (its last-arriving block has said so): the two must never wait for each other.
    // An ordinary launch (first evaluation of an align) never qualifies.
    if (!h->arrive_ctr.p) {
      HIP_TRY(h, h->arrive_ctr.ensure(2));
      HIP_TRY(h, hipMemsetAsync(h->arrive_ctr.p, 0, h->arrive_ctr.cap * sizeof(unsigned int), s));
      HIP_TRY(h, hipStreamSynchronize(s));
      HIP_TRY(h, h->arrived.ensure(2));
      h->arrived.h[0] = h->arrived.h[1] = 0;
    }
    // (The launch in flight has just been given its pose; if it was queued behind its predecessor it is only now
    // starting.  The next kernel is not needed for another ~10 us, so the host can afford to watch the arrival
    // word for a few microseconds before it decides.)
    bool in_flight_resident = false;
    if (h->streams_this_align && via_mailbox) {
      const auto t_arr = std::chrono::steady_clock::now();
      for (unsigned spins = 0;; ++spins) {
        if (__atomic_load_n(&h->arrived.h[buf], __ATOMIC_ACQUIRE) == seq) { in_flight_resident = true; break; }
        if ((spins & 63) == 63 && std::chrono::steady_clock::now() - t_arr > std::chrono::microseconds(6)) break;
        _mm_pause();
      }
    }
    h->pre_on2 = in_flight_resident ? (h->cur_on2 ^ 1) : h->cur_on2;
    if (in_flight_resident) h->n_prelaunch_overlapped++;
    launch_derivatives(px, py, pz, h->n_src, h->geom, h->cell2leaf.p, records, h->cent.p, pc, nullptr, 1, ec, h->partials.p,
                       h->counters.p, d_out, h->pre_on2 ? h->stream2 : s, h->flag.d + (size_t)h->pre_buf * 2 * EV_WORDS,
                       h->pre_seq, h->mbox, xinfo, h->pre_round, h->arrive_ctr.p + h->pre_buf, h->arrived.d + h->pre_buf);
    HIP_TRY(h, hipGetLastError());
  }
  if (dev_out) {
    int rc = h->red.allreduce_device(h->dres.p, EV_WORDS, s, &h->err);
    if (rc) return rc;
    HIP_TRY(h, hipMemcpyAsync(h->result.h, h->dres.p, EV_WORDS * sizeof(double), hipMemcpyDeviceToHost, s));
    if (h->timing && h->ev2) HIP_TRY(h, hipEventRecord(h->ev2, s));
  }
  if (spin) {
    int rc = wait_slots(h, seq, 1, buf);
    if (rc) return rc;
  } else {
    HIP_TRY(h, hipStreamSynchronize(s));
  }
  h->tm.n_eval_launches++;
  if (h->timing) {
    float ms = 0;
    HIP_TRY(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
    h->tm.ms_last_eval_kernel = ms;
    h->tm.ms_eval_kernel_total += ms;
    h->tm.n_timed_evals++;
    if (dev_out && h->ev2) {   // RCCL: the all-reduce and the read-back behind the kernel, device time
      float mr = 0;
      if (hipEventElapsedTime(&mr, h->ev1, h->ev2) == hipSuccess) {
        h->tm.ms_last_reduce_kernel = mr;
        h->tm.ms_reduce_kernel_total += mr;
      } else {
        (void)hipGetLastError();
      }
    }
  }
  double words[EV_WORDS];
  if (spin) {
    for (int v = 0; v < EV_WORDS; ++v) std::memcpy(&words[v], &h->flag.h[((size_t)buf * EV_WORDS + v) * 2 + 1], sizeof(double));
  } else {
    std::memcpy(words, h->result.h, sizeof(words));
  }
  if (p2p) {
    if (words[EV_FAIL] == 3.0) {
      h->n_p2p_host_finishes++;
      // this rank's sum is published, a peer was more than 20 ms late (a starved host over there):
      // the kernel is gone, the host finishes the same exchange -- same rows, same rank order
      int rc = h->red.p2p_finish_on_host(xround, words, EV_WORDS, &h->err);
      if (rc) return rc;
    }
    // (otherwise the words ARE the global sums already)
  } else if (!dev_out) {
    // (kernel timing on: the cross-rank sum's own wall time, from the local sum in hand to the global one -- what the
    // --gpus N bench line reports per transport)
    const auto t_red = std::chrono::steady_clock::now();
    int rc = h->red.allreduce_host(words, EV_WORDS, &h->err);
    if (rc) return rc;
    if (h->timing && h->red.mode() != NDT_REDUCE_NONE) {
      const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_red).count();
      h->tm.ms_last_reduce_kernel = ms;
      h->tm.ms_reduce_kernel_total += ms;
    }
  }
  // word 31 is zero by construction; the in-kernel final sum raises it when it gave up waiting
  // for a partial row (a lost hand-off must not look like a converged result), and the kernel
  // never produces a non-finite score from finite records.  With several summing blocks (SUMMER_SPLIT) only one of them
  // owns word 31: the others say "gave up" by publishing their words as NaN -- read here as a lost row, AFTER the
  // cross-rank sum (a NaN goes through it: every rank sees it and repeats the evaluation).
  if (words[EV_FAIL] == 0.0)
    for (int v = 0; v < EV_WORDS; ++v)
      if (std::isnan(words[v])) { words[EV_FAIL] = 1.0; break; }
  if (words[EV_FAIL] == 2.0 && via_mailbox) {
    // the pre-launched kernel gave up waiting for its pose (this thread was away for > 20 ms):
    // nothing was evaluated.  Evaluate the ordinary way, and stop pre-launching for this align.
    h->n_prelaunch_timeouts++;
    quit_prelaunched(h);
    if (p2p) h->red.p2p_set_round(xround - 1);  // nothing was exchanged under this tag: the re-evaluation uses it
    h->prelaunch_armed = false;  // ordinary launches for the rest of this align; the next align tries again
    // Every block times out on its own clock: block 0 wrote the notice, blocks whose deadline comes microseconds later
    // may still see the late pose and compute -- rows (and tickets) tagged with the abandoned sequence number.  With
    // two streams nothing orders the re-launch behind them: drain both before the rows are reused.
    h->counters_zeroed = 0;
    HIP_TRY(h, hipStreamSynchronize(s));
    HIP_TRY(h, hipStreamSynchronize(h->stream2));
    if (++h->prelaunch_strikes >= 3) h->prelaunch_suspended = true;  // three aligns in a row: a chronically starved host
    return evaluate(h, p, T, need_h, out, score_only);
  }
  if (words[EV_FAIL] == 1.0 && via_mailbox) {
    // Every block of a pre-launched kernel times out on its OWN clock: when the pose lands near the
    // deadline (or the grid is larger than the machine, so that late blocks start after the first
    // wave's 20 ms) some blocks compute while others have left, and the final sum then misses rows.
    // Nothing usable was evaluated -- same remedy as a time-out: an ordinary launch of the same pose.
    h->n_prelaunch_timeouts++;
    quit_prelaunched(h);
    if (p2p) h->red.p2p_set_round(xround - 1);  // a sum that missed rows is never published (sum_rows)
    h->prelaunch_armed = false;
    h->counters_zeroed = 0;  // ticket mode: the partial tickets of the abandoned launch are not zero
    HIP_TRY(h, hipStreamSynchronize(s));  // the abandoned grid has drained before its rows are reused
    HIP_TRY(h, hipStreamSynchronize(h->stream2));
    if (++h->prelaunch_strikes >= 3) h->prelaunch_suspended = true;
    return evaluate(h, p, T, need_h, out, score_only);
  }
  if (words[EV_FAIL] == 1.0 && !via_mailbox && !safe_retry && !dev_out) {
    // An ordinary launch whose summing block gave up waiting for a row (its blocks were not all resident within
    // SUM_TIMEOUT_TICKS: a device shared with other processes).  Nothing usable was evaluated and nothing was
    // exchanged: once more, stream-synchronised, with the final sum made by the block that draws the last ticket --
    // a launch in which no block waits for another.  Only if THAT fails is the evaluation an error.
    h->n_lost_row_retries++;
    quit_prelaunched(h);
    if (p2p) h->red.p2p_set_round(xround - 1);
    h->prelaunch_armed = false;
    h->counters_zeroed = 0;
    HIP_TRY(h, hipStreamSynchronize(s));
    HIP_TRY(h, hipStreamSynchronize(h->stream2));
    return evaluate(h, p, T, need_h, out, score_only, /*safe_retry=*/true);
  }
  if (words[EV_FAIL] != 0.0 || !std::isfinite(words[EV_SCORE])) {
    h->counters_zeroed = 0;  // the ticket words may be stale: re-zero them before the next launch
    return fail(h, NDT_ERR_HIP, words[EV_FAIL] != 0.0 ? "derivative kernel: a partial row never arrived (hand-off lost)"
                                                      : "derivative kernel returned a non-finite score");
  }
  unpack_eval(words, out);
  if (!score_only) finish_eval(h->prm, h->have_reg ? h->reg_pose : nullptr, p, need_h, out);
  return NDT_OK;
}


}  // namespace engine
}  // namespace ndt

extern "C" {

int ndt_align(ndt_handle* h, const float guess[16], ndt_result* out) {
  if (!h || !guess || !out) return NDT_ERR_INVALID_ARG;
  int rc = bind_device(h);
  if (rc) return rc;
  struct Busy {   // (the heartbeat pauses while an align runs, and counts its period from the align's end)
    KeepWarm& k;
    explicit Busy(KeepWarm& kw) : k(kw) { k.touch(); }
    ~Busy() { k.touch(); }
  } busy(h->keepwarm);
  h->spec_build_failed = false;
  h->spec_first = first_eval_behind_build(h);   // the build's verdict is then collected inside the first evaluation
  if (!h->spec_first) {
    rc = ready_for_eval(h);
    if (rc) {
      // the reference returns the prior with converged = false (ref: svn_ndt_impl.hpp:682-702)
      std::memset(out, 0, sizeof(*out));
      std::memcpy(out->final_transformation, guess, sizeof(float) * 16);
      return rc;
    }
    rc = maybe_sort_source(h, guess);
    if (rc) return rc;
  }
  const double dev_ms0 = h->tm.ms_eval_kernel_total;
  // (the moment the FIRST evaluation of this align is in hand: the stream-placement comparison below times what
  // comes after it -- the first evaluation carries the wait for a deferred build or hand-off, 0.06-0.12 ms of a 0.5 ms
  // align, and an align without one would otherwise look 15 % faster than one with: ADVICE r04)
  bool first_done = false;
  std::chrono::steady_clock::time_point t_first_done;
  EvalFn fn = [h, &first_done, &t_first_done](const double* p, const float* T, bool need_h, Eval* e) {
    const int r = evaluate(h, p, T, need_h, e);
    if (!first_done) { first_done = true; t_first_done = std::chrono::steady_clock::now(); }
    return r;
  };
  const int64_t n_total = h->n_src_global >= 0 ? h->n_src_global : (int64_t)h->n_src;
  h->prelaunch_armed = true;
  const int64_t timeouts0 = h->n_prelaunch_timeouts;
  // placement of the waiting kernels for this align (see auto_one_stream)
  const bool auto_mode = h->two_streams && h->prm.prelaunch == NDT_PRELAUNCH_AUTO;
  ++h->n_auto_aligns;   // (probes: the 6th and 14th align of a handle, so that a shared device is noticed early, then every 32nd)
  h->probing = auto_mode && auto_probe_enabled() && (h->n_auto_aligns == 6 || h->n_auto_aligns == 14 || h->n_auto_aligns % 32 == 0);
  h->streams_this_align = auto_mode && (h->auto_one_stream == h->probing);   // two streams unless AUTO settled on one (probe: the other)
  const int64_t used0 = h->n_prelaunch_used, launches0 = h->tm.n_eval_launches;
  rc = newton_align(h->prm, n_total, guess, fn, out, /*hessian_in_trials=*/true, &h->history);
  h->prelaunch_armed = false;
  if (h->spec_first) {   // (no evaluation was asked for: the pending build is still to be collected)
    h->spec_first = false;
    const int rs = ready_for_eval(h);
    if (rs) { h->spec_build_failed = true; rc = rs; }
  }
  if (h->spec_build_failed) {   // as when the build's failure is found before the loop (above)
    h->spec_build_failed = false;
    quit_prelaunched(h);
    std::memset(out, 0, sizeof(*out));
    std::memcpy(out->final_transformation, guess, sizeof(float) * 16);
    return rc;
  }
  if (auto_mode && rc == NDT_OK && h->n_prelaunch_timeouts == timeouts0) {
    const int64_t launched = h->tm.n_eval_launches - launches0;
    if (launched >= 8 && h->n_prelaunch_used - used0 >= launched - 2) {   // a pre-launched align of some length
      const int which = h->streams_this_align ? 0 : 1;
      // per evaluation BEHIND the first one: like with like, whether or not this align had a build to wait for
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_first_done).count() /
                        (double)(launched - 1);
      // the BEST recent sample of a placement, slowly forgotten (+2 % per align): a host hiccup can only make a
      // sample slower, so it can neither inflate the figure a probe is compared with nor pass for a fast probe
      double& m = h->us_eval_mean[which];
      m = m == 0.0 ? us : std::min(m * 1.02, us);
      const int cur = h->auto_one_stream ? 1 : 0, other = 1 - cur;
      if (h->probing && h->us_eval_mean[cur] > 0.0 && us < 0.85 * h->us_eval_mean[cur]) {
        h->auto_one_stream = !h->auto_one_stream;
        h->us_eval_mean[cur] = 0.0;     // the situation has changed: what was measured in it is stale
        (void)other;
        ++h->n_auto_switches;
      }
    }
  }
  h->probing = false;
  if (h->n_prelaunch_timeouts == timeouts0) h->prelaunch_strikes = 0;
  quit_prelaunched(h);  // the kernel enqueued for an evaluation that never came
  out->ms_device = h->tm.ms_eval_kernel_total - dev_ms0;
  return rc;
}

int ndt_get_iteration_history(const ndt_handle* h, float* transforms16, double* transform_probability, double* nvtl, int cap) {
  if (!h || cap < 0) return NDT_ERR_INVALID_ARG;
  const int n = (int)h->history.size();
  const int m = n < cap ? n : cap;
  if (transforms16 && m) std::memcpy(transforms16, h->history.transforms.data(), (size_t)m * 16 * sizeof(float));
  if (transform_probability && m) std::memcpy(transform_probability, h->history.transform_probability.data(), (size_t)m * sizeof(double));
  if (nvtl && m) std::memcpy(nvtl, h->history.nvtl.data(), (size_t)m * sizeof(double));
  return n;
}

int ndt_score_transform(ndt_handle* h, const float T[16], ndt_score* out) {
  if (!h || !T || !out) return NDT_ERR_INVALID_ARG;
  std::memset(out, 0, sizeof(*out));
  int rc = bind_device(h);
  if (rc) return rc;
  rc = ready_for_eval(h);
  if (rc) return rc;
  double p[6];
  matrix_to_pose(T, p);  // only feeds the (unused) angle tables
  Eval e;
  rc = evaluate(h, p, T, false, &e, /*score_only=*/true);
  if (rc) return rc;
  const int64_t n_total = h->n_src_global >= 0 ? h->n_src_global : (int64_t)h->n_src;
  out->score = e.score;
  out->transform_probability = n_total > 0 ? e.score / (double)n_total : 0.0;
  out->nearest_voxel_transformation_likelihood = e.n_with > 0 ? e.nvtl_sum / e.n_with : 0.0;
  out->n_pairs = (int64_t)e.n_pairs;
  out->n_points_with_neighbors = (int64_t)e.n_with;
  return NDT_OK;
}

int ndt_comm_info(char* path_buf, size_t cap) { return Reducer::library_info(path_buf, cap); }

int ndt_comm_rank_count(const ndt_handle* h) { return h ? h->red.rank_count() : NDT_ERR_INVALID_ARG; }

// overlap: host work that does not need this evaluation's results, run between the launch and the wait
static int eval_batch(ndt_handle* h, const double* poses6, const float* transforms, int K, int compute_hessian,
                      bool score_only, double* out, void (*overlap)(void*) = nullptr, void* overlap_ctx = nullptr) {
  if (!h || !poses6 || !out || K <= 0) return NDT_ERR_INVALID_ARG;
  int rc = bind_device(h);
  if (rc) return rc;
  rc = ready_for_eval(h);
  if (rc) return rc;
  {  // source ordering, by the first pose of the batch (the particles of an SVN iteration are close)
    float T0[16];
    if (!transforms) pose_to_matrix(poses6, T0);
    rc = maybe_sort_source(h, transforms ? transforms : T0);
    if (rc) return rc;
  }
  hipStream_t s = h->stream;
  HIP_TRY(h, h->hposes.ensure((size_t)K));
  HIP_TRY(h, h->dposes.ensure((size_t)K));
  // Fast hand-off of a batch (one SVN iteration): the poses are written by the host straight into
  // BAR-mapped device memory (no H2D copy launch) and every pose's result comes back as 32 tagged
  // slots in pinned host memory that this thread polls (no D2H copy launch, no stream sync).
  const bool dev_red = h->red.wants_device_buffer();
  bool fast = !dev_red && !h->timing && h->prm.wait_mode == NDT_WAIT_SPIN && ensure_mailbox(h);
  if (fast && (size_t)K > h->bposes_cap) {
    if (h->bposes) (void)hipFree(h->bposes);
    h->bposes = nullptr;
    h->bposes_cap = 0;
    void* p = nullptr;
    const size_t want = (size_t)K + 8;
    if (hipExtMallocWithFlags(&p, want * sizeof(PoseConsts), hipDeviceMallocFinegrained) == hipSuccess && p) {
      h->bposes = static_cast<PoseConsts*>(p);
      h->bposes_cap = want;
    } else {
      (void)hipGetLastError();
      fast = false;
    }
  }
  PoseConsts* stage = fast ? h->bposes : h->hposes.h;
  for (int k = 0; k < K; ++k) {
    float T[16];
    const float* Tk = transforms ? transforms + 16 * (size_t)k : T;
    if (!transforms) pose_to_matrix(poses6 + 6 * (size_t)k, T);
    PoseConsts pc;
    fill_pose_consts(poses6 + 6 * (size_t)k, Tk, &pc);
    std::memcpy(&stage[k], &pc, sizeof(pc));
    if (k == 0) h->hposes.h[0] = pc;
  }
  if (fast) _mm_sfence();  // the write-combined BAR stores are on their way before the doorbell rings
  EvalConsts ec = make_eval_consts(h, compute_hessian != 0);
  ec.score_only = score_only ? 1 : 0;
  const VoxelRecord* records = nullptr;
  rc = records_for_eval(h, &ec, &records);
  if (rc) return rc;
  rc = ensure_partials(h, derivs_partials_words(h->n_src, K, h->n_cus));
  if (rc) return rc;
  HIP_TRY(h, h->result.ensure((size_t)K * EV_WORDS));
  HIP_TRY(h, h->dres.ensure((size_t)K * EV_WORDS));
  rc = ensure_counters(h, (size_t)K * derivs_counters_per_pose());
  if (rc) return rc;
  if (fast) {
    rc = ensure_flag_slots(h, (size_t)K);
    if (rc) return rc;
  } else {
    HIP_TRY(h, hipMemcpyAsync(h->dposes.p, h->hposes.h, (size_t)K * sizeof(PoseConsts), hipMemcpyHostToDevice, s));
  }
  const unsigned long long seq = g_launch_seq.fetch_add(1, std::memory_order_relaxed);
  const bool bracket = h->timing && timing_brackets_launch();
  if (bracket) HIP_TRY(h, hipEventRecord(h->ev0, s));
  launch_derivatives(h->src_sorted ? h->ox.p : h->vx, h->src_sorted ? h->oy.p : h->vy,
                     h->src_sorted ? h->oz.p : h->vz, h->n_src, h->geom, h->cell2leaf.p, records, h->cent.p,
                     h->hposes.h[0], fast ? h->bposes : h->dposes.p, K, ec, h->partials.p, h->counters.p, h->dres.p, s,
                     fast ? h->flag.d : nullptr, seq, nullptr, nullptr, 0ull, nullptr, nullptr,
                     h->timing && !bracket ? h->ev0 : nullptr, h->timing && !bracket ? h->ev1 : nullptr);
  HIP_TRY(h, hipGetLastError());
  if (bracket) HIP_TRY(h, hipEventRecord(h->ev1, s));
  if (overlap) overlap(overlap_ctx);
  if (fast) {
    rc = wait_slots(h, seq, K);
    if (rc) return rc;
    for (int k = 0; k < K; ++k)
      for (int v = 0; v < EV_WORDS; ++v)
        std::memcpy(&h->result.h[(size_t)k * EV_WORDS + v], &h->flag.h[((size_t)k * EV_WORDS + v) * 2 + 1], sizeof(double));
  } else {
    if (dev_red) {
      rc = h->red.allreduce_device(h->dres.p, K * EV_WORDS, s, &h->err);
      if (rc) return rc;
    }
    HIP_TRY(h, hipMemcpyAsync(h->result.h, h->dres.p, (size_t)K * EV_WORDS * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipStreamSynchronize(s));
  }
  h->tm.n_eval_launches++;
  if (h->timing) {
    float ms = 0;
    HIP_TRY(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
    h->tm.ms_last_eval_kernel = ms;
    h->tm.ms_eval_kernel_total += ms;
    h->tm.n_timed_evals++;
  }
  for (int k = 0; k < K; ++k)
    if (h->result.h[(size_t)k * EV_WORDS + EV_FAIL] != 0.0 || !std::isfinite(h->result.h[(size_t)k * EV_WORDS + EV_SCORE])) {
      h->counters_zeroed = 0;
      return fail(h, NDT_ERR_HIP, "derivative kernel (batched): a partial row never arrived or the score is not finite");
    }
  std::memcpy(out, h->result.h, (size_t)K * EV_WORDS * sizeof(double));
  if (!h->red.wants_device_buffer() && h->red.mode() != NDT_REDUCE_NONE) {
    rc = h->red.allreduce_host_batch(out, K, &h->err);   // (P2P: one exchange round per 64 poses, not one per pose)
    if (rc) return rc;
  }
  if (score_only) return NDT_OK;
  // ridge / regularisation / guards, then repack so callers see finished values
  for (int k = 0; k < K; ++k) {
    double* w = out + (size_t)k * EV_WORDS;
    Eval e;
    unpack_eval(w, &e);
    finish_eval(h->prm, h->have_reg ? h->reg_pose : nullptr, poses6 + 6 * (size_t)k, compute_hessian != 0, &e);
    w[EV_SCORE] = e.score;
    for (int i = 0; i < 6; ++i) w[EV_G + i] = e.g[i];
    int idx = EV_H;
    for (int i = 0; i < 6; ++i)
      for (int j = i; j < 6; ++j) w[idx++] = e.H[6 * i + j];
  }
  return NDT_OK;
}

int ndt_eval_derivatives(ndt_handle* h, const double* poses6, const float* transforms, int K,
                         int compute_hessian, double* out) {
  return eval_batch(h, poses6, transforms, K, compute_hessian, false, out);
}

// internal (ndt_svn.cpp; not in the public header): ndt_eval_derivatives with host work run while the
// batched kernel is in flight
int ndt_eval_derivatives_overlapped(ndt_handle* h, const double* poses6, const float* transforms, int K,
                                    int compute_hessian, double* out, void (*overlap)(void*), void* ctx) {
  return eval_batch(h, poses6, transforms, K, compute_hessian, false, out, overlap, ctx);
}

int ndt_score_transforms(ndt_handle* h, const float* transforms, int K, ndt_score* out) {
  if (!h || !transforms || !out || K <= 0) return NDT_ERR_INVALID_ARG;
  std::vector<double> poses6(6 * (size_t)K, 0.0), words((size_t)K * EV_WORDS);  // the angle tables are not used
  int rc = eval_batch(h, poses6.data(), transforms, K, 0, true, words.data());
  if (rc) return rc;
  const int64_t n_total = h->n_src_global >= 0 ? h->n_src_global : (int64_t)h->n_src;
  for (int k = 0; k < K; ++k) {
    const double* w = &words[(size_t)k * EV_WORDS];
    out[k].score = w[EV_SCORE];
    out[k].transform_probability = n_total > 0 ? w[EV_SCORE] / (double)n_total : 0.0;
    out[k].nearest_voxel_transformation_likelihood = w[EV_NWITH] > 0 ? w[EV_NVTL] / w[EV_NWITH] : 0.0;
    out[k].n_pairs = (int64_t)w[EV_NPAIRS];
    out[k].n_points_with_neighbors = (int64_t)w[EV_NWITH];
  }
  return NDT_OK;
}

void ndt_unpack_eval(const double* w, double* score, double* g6, double* H36) {
  Eval e;
  unpack_eval(w, &e);
  if (score) *score = e.score;
  if (g6) std::memcpy(g6, e.g, sizeof(e.g));
  if (H36) std::memcpy(H36, e.H, sizeof(e.H));
}

int ndt_angle_tables(const double pose6[6], float j_ang[24], float h_ang[45]) {
  if (!pose6 || !j_ang || !h_ang) return NDT_ERR_INVALID_ARG;
  angle_tables(pose6, j_ang, h_ang);
  return NDT_OK;
}

int ndt_gauss_constants(double resolution, double outlier_ratio, double* d1, double* d2) {
  if (!d1 || !d2 || !(resolution > 0.0)) return NDT_ERR_INVALID_ARG;
  gauss_constants(resolution, outlier_ratio, d1, d2);
  return NDT_OK;
}

int ndt_transform_source(ndt_handle* h, const float T[16], float* out_xyz, size_t cap_points) {
  if (!h || !T || !out_xyz) return NDT_ERR_INVALID_ARG;
  int rc = bind_device(h);
  if (rc) return rc;
  if (cap_points < h->n_src) return fail(h, NDT_ERR_INVALID_ARG, "output buffer too small");
  if (h->n_src == 0) return NDT_OK;
  rc = settle_source(h);
  if (rc) return rc;
  PoseConsts pc{};
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) pc.R[3 * i + j] = T[4 * j + i];
    pc.t[i] = T[12 + i];
  }
  DevBuf<float> tmp;
  HIP_TRY(h, tmp.ensure(3 * h->n_src));
  launch_transform(h->vx, h->vy, h->vz, h->n_src, pc, tmp.p, h->stream);
  hipError_t e = hipMemcpyAsync(out_xyz, tmp.p, 3 * h->n_src * sizeof(float), hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  tmp.release();
  HIP_TRY(h, e);
  return NDT_OK;
}

int ndt_newton_align(const ndt_params* p, int64_t n_source_total, const float guess[16],
                     const float* reg_pose, ndt_eval_fn fn, void* ctx, ndt_result* out) {
  if (!p || !guess || !fn || !out) return NDT_ERR_INVALID_ARG;
  const ndt_params prm = *p;
  EvalFn wrap = [&](const double* pose, const float* T, bool need_h, Eval* e) -> int {
    double words[NDT_EVAL_WORDS];
    std::memset(words, 0, sizeof(words));
    int rc = fn(ctx, pose, T, need_h ? 1 : 0, words);
    if (rc) return rc;
    unpack_eval(words, e);
    finish_eval(prm, reg_pose, pose, need_h, e);
    return 0;
  };
  return newton_align(prm, n_source_total, guess, wrap, out);
}


}  // extern "C"
