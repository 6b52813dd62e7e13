// ndt_api.hip -- the C-ABI of include/ndt_hip.h on top of the HIP kernels.
//
// One handle = one engine instance = one HIP stream on one gfx950 device; it
// owns every device allocation.  There is no CPU path: without a device every
// compute call fails with NDT_ERR_NO_DEVICE.
#include <immintrin.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <random>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <limits>
#include <string>
#include <thread>
#include <map>
#include <unordered_map>
#include <vector>

#include "../../include/ndt_hip.h"
#include "ndt_comm.h"
#include "ndt_keepwarm.h"
#include "ndt_kernels.h"
#include "ndt_newton.h"
#include "ndt_repack_pool.h"
#include "ndt_tuning.h"

using namespace ndt;

namespace {

template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t cap = 0;
  hipError_t ensure(size_t n) {
    if (n <= cap) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    size_t want = n + n / 8 + 64;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), want * sizeof(T));
    if (e == hipSuccess) cap = want;
    return e;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
};

// The dense cell -> leaf index with four readable ints in front of the first cell and behind the last: the 27-cell
// neighbourhoods load a row of three x-adjacent cells with ONE 12-byte load, which at the ends of the grid starts one
// or two ints outside it (those lanes are masked, the bytes only have to be mapped).  Same member names as DevBuf.
struct IndexGrid {
  DevBuf<int> raw;
  int* p = nullptr;
  size_t cap = 0;
  hipError_t ensure(size_t n) {
    if (n <= cap) return hipSuccess;
    p = nullptr;
    cap = 0;
    hipError_t e = raw.ensure(n + 8);
    if (e == hipSuccess) { p = raw.p + 4; cap = raw.cap - 8; }
    return e;
  }
  void release() { raw.release(); p = nullptr; cap = 0; }
};

template <typename T>
struct PinBuf {  // pinned, device-mapped host memory
  T* h = nullptr;
  T* d = nullptr;
  size_t cap = 0;
  hipError_t ensure(size_t n) {
    if (n <= cap) return hipSuccess;
    release();
    hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&h), n * sizeof(T), hipHostMallocMapped);
    if (e != hipSuccess) { h = nullptr; return e; }
    e = hipHostGetDevicePointer(reinterpret_cast<void**>(&d), h, 0);
    if (e != hipSuccess) { (void)hipHostFree(h); h = nullptr; return e; }
    cap = n;
    return hipSuccess;
  }
  void release() {
    if (h) (void)hipHostFree(h);
    h = nullptr; d = nullptr; cap = 0;
  }
};

}  // namespace

struct ndt_handle {
  ndt_params prm;
  int device = -1;
  KeepWarm keepwarm;                   // optional idle-time heartbeat (ndt_set_keepwarm), off by default
  hipStream_t stream = nullptr;
  hipStream_t stream2 = nullptr;       // pre-launched evaluation kernels alternate between `stream` and this one
  hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr;
  std::string err;

  // target
  DevBuf<float> tx, ty, tz;          // owned copy when the target came from the host
  size_t n_tgt = 0;
  bool have_grid = false;
  GridGeom geom{};
  int max_b[3] = {0, 0, 0};
  DevBuf<uint32_t> keys, vals, keys2, vals2;
  DevBuf<char> sort_tmp;
  DevBuf<int> nleaf;                 // [0] slots, [1] valid
  DevBuf<int> leaf_start, leaf_cnt, run_counts, run_offsets, fin_counts;
  std::unique_ptr<RepackPool> pool;   // repack workers of the host hand-off, created on first use
  DevBuf<uint32_t> sort_tags;         // tagged tile counts of the fused sort passes
  uint32_t sort_seq = 0;              // ... and their launch tag counter
  DevBuf<uint32_t> run_tags;          // tagged block leaf counts of the fused run search
  uint32_t run_seq = 0;
  long long n_fused_sort_fallbacks = 0;
  DevBuf<int> bucket_off;             // first point of every bucket (two-launch bucketed build)
  DevBuf<int> bnd;                    // its 8 bounds words {min xyz, max xyz, #finite, largest bucket}; neutral between builds
  long long n_bucket_builds = 0, n_bucket_fallbacks = 0;
  int bucket_skip = 0, bucket_backoff = 0;  // builds to go before the two-launch build is tried again after a decline
  const float* vx = nullptr;          // the source as evaluated: the engine's own copy (sx/sy/sz) or,
  const float* vy = nullptr;          // after ndt_set_source_device_view, the caller's arrays
  const float* vz = nullptr;
  unsigned int build_seq = 0;         // tag of the build whose completion the host polls for
  int n_cus = 0;                      // compute units of the device (a fused sort pass needs one per tile)
  DevBuf<double> leaf_sums;
  DevBuf<int> brows;                 // per-block bounds rows
  DevBuf<unsigned int> tickets;      // [0] bounds, [1] run-count, [2] finalize kernel, [4..5] the two-launch build's 64-bit tail word; zero between launches
  DevBuf<BuildGeom> gd;              // geometry + sort plan of the build, derived on the device
  PinBuf<BuildGeom> gdh;             // ... and its host-visible copy
  DevBuf<float> xyz4;                // packed float4 copy of the target for the gather
  IndexGrid cell2leaf;
  DevBuf<VoxelRecord> rec;
  // the same table as 48-byte PackedRecords (ndt_set_record_format): written behind every build while the packed
  // format is selected, or on demand when it is selected afterwards
  DevBuf<PackedRecord> prec;
  bool prec_valid = false;
  int record_format = NDT_RECORDS_F64;
  DevBuf<float> cent;                // 4 floats per leaf slot: f32 centroid (what the radius search tests) + chain link
  DevBuf<LeafStats> stats;
  int n_slots = 0, n_valid = 0;
  // The dense index grid is filled with -1 once per allocation; afterwards only the cells the
  // previous build published are reset (a 33 MB fill per build otherwise).
  size_t grid_clean_cap = 0;  // capacity that is -1 everywhere except the dirty leaves' cells
  int grid_dirty_slots = 0;   // slots of `stats` whose cells may hold an index
  double ms_build = 0;

  // source
  DevBuf<float> sx, sy, sz;
  size_t n_src = 0;
  // the same points in block order of the target grid (see sort_source_by_blocks); valid until
  // the source or the target changes
  DevBuf<float> ox, oy, oz;
  DevBuf<uint32_t> skeys, skeys2, svals, svals2;
  DevBuf<char> ssort_tmp;
  DevBuf<BuildGeom> splan;
  bool src_sorted = false;
  int64_t n_src_global = -1;

  // staging
  // Host hand-off (ndt_set_target / ndt_set_source and their SoA forms): two lanes, each with its own pinned staging
  // buffer and stream, so that the target's transfer and build run while the source is repacked.
  struct UploadLane {
    PinBuf<float> stage;             // pinned, device-mapped staging: the cloud chunk by chunk as [x | y | z]
    hipEvent_t done = nullptr;       // the last pull kernel out of `stage` has finished
    hipEvent_t t0 = nullptr, t1 = nullptr;  // kernel timing: before the first copy / behind the last
    bool busy = false;               // `done` has been recorded and not yet waited for
    bool timed = false;              // t0 / t1 were recorded for the hand-off in flight
    ndt_handoff_lane_timing tm{};    // breakdown of the lane's last hand-off
  };
  UploadLane lane_t, lane_s;
  hipStream_t ustream = nullptr;     // the source lane's stream (the target lane shares `stream` with the build)
  bool src_upload_pending = false;   // the engine's streams have not yet been ordered behind the source hand-off
  int handoff_mode = NDT_HANDOFF_ASYNC;
  // The voxel-grid build of an asynchronous hand-off: enqueued by ndt_set_target, its verdict collected by the first
  // call that needs the grid (settle()).
  struct BuildRun {
    const float* x = nullptr;
    const float* y = nullptr;
    const float* z = nullptr;
    size_t n = 0;
    int dirty_slots = 0;
    size_t clean_cap = 0;
    bool fused = false, fused_sort = false, bucketed_ok = false;
    int min_pts = 0, max_leaves = 0;
    float leaf = 0, inv_leaf = 0;
    bool build_events = false, poll_done = false;
    std::chrono::steady_clock::time_point t0;
    // the attempt in flight
    int attempt = 0;
    bool optimistic = false, bucketed = false;
    int done_tag = 0;
  };
  BuildRun brun;
  bool build_pending = false;
  int deferred_rc = 0;               // status of a deferred build that failed, until a call that needs the grid (or ndt_wait) has reported it
  std::string deferred_msg;
  double ms_settle_wait = 0;         // time the collecting call waited for the pending build's verdict
  PinBuf<double> result;             // evaluation results (K * EV_WORDS)
  PinBuf<int> small;                 // bounds / counters read-back
  PinBuf<unsigned long long> flag;   // 32 result slots {seq, value} the single-pose kernel writes for the host
  DevBuf<double> partials, dres;
  DevBuf<unsigned int> counters;     // per-pose tickets of the in-kernel final reduction
  size_t counters_zeroed = 0;
  DevBuf<PoseConsts> dposes;
  PinBuf<PoseConsts> hposes;
  PoseConsts* bposes = nullptr;      // pose batch in BAR-mapped fine-grained device memory (host writes it directly)
  size_t bposes_cap = 0;
  size_t flag_slots = 0;             // poses the pinned result slots `flag` can hold

  // device-resident keyframe archive (pointsArchive of the drivers, ref: run/pipeline.cpp:784)
  struct Keyframe {
    DevBuf<float> x, y, z;
    size_t n = 0;
  };
  std::unordered_map<int64_t, Keyframe> keyframes;
  // buffers of erased keyframes, reused by the next ndt_keyframe_put (a sliding window erases one keyframe and archives
  // one per scan: hipMalloc / hipFree of three arrays each cost more than the upload they frame); at most 4 are kept
  std::vector<Keyframe> keyframe_pool;

  // multi-grid target [RECALLED] (tier4 MultiGridNormalDistributionsTransform): the valid leaves of
  // every separately voxelised cloud, on the host (adding a map tile is not a per-scan operation);
  // ndt_multigrid_create_kdtree assembles their union into the device table
  struct MultiGridEntry {
    std::vector<VoxelRecord> rec;
    std::vector<LeafStats> stats;
    std::vector<int> ijk;  // absolute lattice index of every leaf, 3 ints
    size_t n_points = 0;
    float resolution = 0.0f;
    int min_points = 0, cov_mode = 0;
    double eig_ratio = 0.0;
  };
  std::map<int64_t, MultiGridEntry> mgrids;
  bool multi_active = false;          // the device table is the union (neighbourhood: radius search, chained cells)
  std::vector<LeafStats> multi_stats; // ... and its leaves in table order, for export

  // pre-launched evaluation (ndt_prelaunch): mailbox in BAR-mapped fine-grained device memory
  PoseMailbox* mbox = nullptr;
  bool mbox_tried = false;
  bool mbox_tagged = true;            // the pose is published as tagged 8-byte granules
  bool mbox_preload = false;          // the waiting kernel fetches its points before the pose arrives (measured: no gain)
  bool prelaunch_armed = false;       // inside ndt_align
  unsigned long long pre_seq = 0;     // sequence number of the kernel that is waiting, 0 = none
  unsigned long long pre_round = 0;   // ... and the cross-rank round it will exchange under (NDT_REDUCE_P2P)
  int pre_on2 = 0;                    // ... and the stream it is on (0: stream, 1: stream2)
  int cur_on2 = 0;                    // stream of the evaluation in flight
  bool two_streams = true;            // NDT_PRELAUNCH_STREAMS != 1
  // NDT_PRELAUNCH_AUTO decides between the two-stream and the one-stream placement of the waiting kernel BY MEASUREMENT:
  // a waiting kernel on the other stream holds its compute units for the whole evaluation of its predecessor -- harmless
  // on a device the engine has to itself, ruinous when another engine's kernels need those units (two ranks on one
  // device: 0.97 against 0.59 ms per step, HISTORY section 5).  The handle keeps the best recent wall time per
  // launched evaluation in the placement in use and runs the 6th, the 14th and then every 32nd align in the other one as a probe; a probe that is
  // 15 % faster switches the handle over (and the probing goes on from there, so it can switch back).
  bool auto_one_stream = false;       // the placement AUTO currently uses
  bool probing = false;               // this align runs in the other placement
  bool streams_this_align = true;     // two-stream placement in effect for the align in flight
  double us_eval_mean[2] = {0.0, 0.0};  // best recent wall time per launched evaluation: [0] two streams, [1] one stream; 0 = no sample yet
  int64_t n_auto_aligns = 0, n_auto_switches = 0;
  DevBuf<unsigned int> arrive_ctr;    // [2] blocks of a pre-launched launch that have started (per result buffer)
  PinBuf<unsigned long long> arrived; // [2] sequence number of the launch whose blocks are all resident
  int pre_buf = 0;                    // ... and the result buffer (0 / 1) it will write
  int prelaunch_strikes = 0;          // consecutive aligns in which a waiting kernel gave up
  bool prelaunch_suspended = false;   // three such aligns in a row (a chronically starved host): no more pre-launching
                                      // on this handle until ndt_set_params is called -- the caller's ndt_params
                                      // are never rewritten
  int flag_toggle = 0;                // result buffer of the latest single-pose launch
  bool pre_need_h = false;
  int64_t n_prelaunch_used = 0, n_prelaunch_quit = 0, n_prelaunch_timeouts = 0;
  int64_t n_lost_row_retries = 0;      // evaluations repeated through the ticketed final sum after a row was lost
  // The first evaluation of an align that follows a DEFERRED build is enqueued behind that build, before its verdict is
  // known (evaluate()): spec_first is set by ndt_align for its first evaluate() call.
  bool spec_first = false;
  bool spec_enabled = true;            // (ndt_debug_set_speculation: in-process A/B)
  int prev_n_valid = 0;                // valid voxels of the grid the build in flight replaces
  bool spec_build_failed = false;      // the deferred build's failure surfaced inside that first evaluation
  int64_t n_spec_used = 0, n_spec_discarded = 0;
  int64_t n_p2p_host_finishes = 0;     // peer-write evaluations whose exchange the host finished (a peer was late)
  int64_t n_prelaunch_overlapped = 0; // pre-launches that went to the other stream (resident before their predecessor ended)

  bool have_reg = false;
  float reg_pose[16];
  IterHistory history;                // per-iteration transforms / scores of the last ndt_align (ndt_get_iteration_history)

  Reducer red;

  bool timing = false;
  ndt_timing tm{};
};

namespace {

int fail(ndt_handle* h, int code, const std::string& msg) {
  if (h) h->err = msg;
  return code;
}

#define HIP_TRY(h, expr)                                                                  \
  do {                                                                                    \
    hipError_t e__ = (expr);                                                              \
    if (e__ != hipSuccess)                                                                \
      return fail(h, (e__ == hipErrorOutOfMemory) ? NDT_ERR_ALLOC : NDT_ERR_HIP,          \
                  std::string(#expr) + ": " + hipGetErrorString(e__));                    \
  } while (0)

int bind_device(ndt_handle* h) {
  HIP_TRY(h, hipSetDevice(h->device));
  return NDT_OK;
}

bool params_valid(const ndt_params* p, std::string* why) {
  if (!(p->resolution > 1e-6f)) { *why = "resolution must be positive"; return false; }
  if (p->search_method != NDT_DIRECT7 && p->search_method != NDT_DIRECT1 && p->search_method != NDT_KDTREE &&
      p->search_method != NDT_DIRECT26) {
    *why = "unknown search method (KDTREE / DIRECT26 / DIRECT7 / DIRECT1)";
    return false;
  }
  if (p->wait_mode != NDT_WAIT_SPIN && p->wait_mode != NDT_WAIT_BLOCK) { *why = "unknown wait_mode"; return false; }
  if (p->source_order < NDT_SOURCE_ORDER_AUTO || p->source_order > NDT_SOURCE_ORDER_SORT) { *why = "unknown source_order"; return false; }
  if (p->prelaunch != NDT_PRELAUNCH_AUTO && p->prelaunch != NDT_PRELAUNCH_OFF && p->prelaunch != NDT_PRELAUNCH_ONE_STREAM) {
    *why = "unknown prelaunch";
    return false;
  }
  if (!(p->outlier_ratio >= 0.0 && p->outlier_ratio < 1.0)) { *why = "outlier_ratio must be in [0,1)"; return false; }
  if (p->max_iterations < 0) { *why = "max_iterations must be >= 0"; return false; }
  return true;
}

// ---- host hand-off ---------------------------------------------------------------------------------
// host AoS / SoA -> device SoA through a lane's pinned, device-mapped staging buffer (ref: run/pipeline.cpp:554-561
// hands host pcl::PointCloud<PointXYZI> clouds).  The cloud is repacked piece by piece by a few persistent host
// threads AND the calling thread (ndt_repack_pool.h) into pinned memory, chunk-major; every finished chunk is pulled
// over PCIe at once by a small kernel of its own that writes the three SoA arrays (launch_pull_chunk), so the
// transfer runs under the repack.  When the function returns the CALLER'S memory has been consumed -- the pull
// kernels and whatever the caller enqueues behind them on `stream` may still be running (`sync` = false); the
// lane's `done` event guards the staging buffer's reuse.
int lane_wait(ndt_handle* h, ndt_handle::UploadLane& lane) {
  if (!lane.busy) return NDT_OK;
  HIP_TRY(h, hipEventSynchronize(lane.done));
  lane.busy = false;
  if (lane.timed) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, lane.t0, lane.t1) == hipSuccess) {
      lane.tm.ms_dma = ms;
      lane.tm.dma_gb_per_s = ms > 0 ? (double)lane.tm.bytes_dma / (ms * 1e-3) / 1e9 : 0.0;
    }
    lane.timed = false;
  }
  return NDT_OK;
}

int upload_soa(ndt_handle* h, ndt_handle::UploadLane& lane, hipStream_t stream, const float* xyz, const float* x,
               const float* y, const float* z, size_t n, size_t stride, DevBuf<float>& dx, DevBuf<float>& dy,
               DevBuf<float>& dz, bool sync) {
  const auto t_begin = std::chrono::steady_clock::now();
  int rc = lane_wait(h, lane);  // the previous hand-off's copies out of this lane's staging buffer
  if (rc) return rc;
  HIP_TRY(h, dx.ensure(n));
  HIP_TRY(h, dy.ensure(n));
  HIP_TRY(h, dz.ensure(n));
  lane.tm = ndt_handoff_lane_timing{};
  if (n == 0) return NDT_OK;
  {  // (grown with slack: a stream of slightly growing clouds must not re-pin its staging buffer scan after scan)
    const size_t need = StageJob::stage_floats(n);
    if (need > lane.stage.cap) HIP_TRY(h, lane.stage.ensure(need + need / 8 + 4096));
  }
  if (!lane.done) {
    HIP_TRY(h, hipEventCreateWithFlags(&lane.done, hipEventDisableTiming));
    HIP_TRY(h, hipEventCreate(&lane.t0));
    HIP_TRY(h, hipEventCreate(&lane.t1));
  }
  StageJob job;
  if (xyz) { job.aos = reinterpret_cast<const char*>(xyz); job.stride = stride; }
  else { job.x = x; job.y = y; job.z = z; }
  job.n = n;
  job.stage = lane.stage.h;
  // a cloud of up to two pieces is not worth a hand-shake with the workers
  const unsigned workers = n > 2 * job.piece ? repack_workers() : 0u;
  if (workers && !h->pool) h->pool.reset(new RepackPool());
  lane.timed = h->timing;
  if (lane.timed) HIP_TRY(h, hipEventRecord(lane.t0, stream));
  // every finished chunk is pulled over PCIe by a small kernel of its own, straight into the SoA arrays
  const float* stage_dev = lane.stage.d;
  stage_cloud(h->pool.get(), workers, job, [&](size_t c, size_t lo, size_t hi) {
    launch_pull_chunk(stage_dev + 3 * lo, hi - lo, job.seg(c), dx.p + lo, dy.p + lo, dz.p + lo, stream);
  });
  HIP_TRY(h, hipGetLastError());
  if (lane.timed) HIP_TRY(h, hipEventRecord(lane.t1, stream));
  HIP_TRY(h, hipEventRecord(lane.done, stream));
  lane.busy = true;
  lane.tm.n_points = (int64_t)n;
  lane.tm.bytes_in = (int64_t)(xyz ? n * stride : 3 * n * sizeof(float));
  lane.tm.bytes_dma = (int64_t)(3 * n * sizeof(float));
  lane.tm.threads = (int)workers + 1;
  lane.tm.ms_repack = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
  if (sync) {
    HIP_TRY(h, hipStreamSynchronize(stream));
    return lane_wait(h, lane);
  }
  return NDT_OK;
}

// The voxel-grid build proper; x/y/z are device pointers.
//
// Steady state ("optimistic"): the dense grid and every scratch buffer exist from an earlier
// build, so the whole chain of launches is enqueued without waiting for the bounds -- the
// geometry and the sort plan are derived on the device (BuildGeom) and checked against the
// capacities the host assumed; ONE synchronisation at the end.  First build, or a cloud that
// outgrew the buffers: the host waits for the geometry once, allocates, and goes on (as
// before).  A refused optimistic build (BG_CAPACITY / BG_PASSES) is repeated that way.
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

// (re)initialise the bounds words of the two-launch build
int neutral_bounds(ndt_handle* h) {
  int w[8];
  bucket_bounds_neutral(w);
  HIP_TRY(h, hipMemcpyAsync(h->bnd.p, w, sizeof(w), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));   // (w lives on this frame)
  return NDT_OK;
}

// everything of a build that does not depend on the attempt: state reset, allocations, constants
int build_begin(ndt_handle* h, const float* x, const float* y, const float* z, size_t n, ndt_handle::BuildRun& br) {
  h->prev_n_valid = h->have_grid && !h->multi_active ? h->n_valid : 0;   // (first_eval_behind_build's size guess)
  h->have_grid = false;
  h->multi_active = false;
  h->src_sorted = false;
  h->n_tgt = n;
  h->n_slots = h->n_valid = 0;
  br = ndt_handle::BuildRun{};
  br.x = x; br.y = y; br.z = z; br.n = n;
  br.dirty_slots = h->grid_dirty_slots;
  br.clean_cap = h->grid_clean_cap;
  h->grid_clean_cap = 0;  // pessimistic until this build has gone through
  h->grid_dirty_slots = 0;
  if (n == 0) return fail(h, NDT_ERR_NO_TARGET, "empty target cloud");
  if (n > (size_t)std::numeric_limits<int>::max() / 2) return fail(h, NDT_ERR_INVALID_ARG, "target too large");
  hipStream_t s = h->stream;
  const int nrows = bounds_rows(n);
  br.min_pts = std::max(3, h->prm.min_points_per_voxel);  // ref: voxel_grid_covariance.h:176-184
  br.max_leaves = (int)(n / (size_t)br.min_pts) + 1;
  const int max_leaves = br.max_leaves;
  br.leaf = h->prm.resolution;
  br.inv_leaf = 1.0f / h->prm.resolution;

  HIP_TRY(h, h->small.ensure(16));
  HIP_TRY(h, h->brows.ensure(8 * (size_t)std::max(nrows, bucket_build_tiles(n))));
  HIP_TRY(h, h->bucket_off.ensure(260));
  if (!h->bnd.p) {
    HIP_TRY(h, h->bnd.ensure(8));
    int rc = neutral_bounds(h);
    if (rc) return rc;
  }
  HIP_TRY(h, h->gd.ensure(1));
  HIP_TRY(h, h->gdh.ensure(1));
  if (!h->tickets.p) {
    HIP_TRY(h, h->tickets.ensure(6));
    HIP_TRY(h, hipMemsetAsync(h->tickets.p, 0, h->tickets.cap * sizeof(unsigned int), s));
  }
  HIP_TRY(h, h->nleaf.ensure(4));  // [0] slots, [1] valid, [2] buckets that declined (two-launch build)
  // a re-allocation of `stats` would lose the cells the previous build published
  if ((size_t)max_leaves > h->stats.cap) br.clean_cap = 0;
  HIP_TRY(h, h->keys.ensure(n));
  HIP_TRY(h, h->xyz4.ensure(4 * n));
  HIP_TRY(h, h->vals.ensure(n));
  HIP_TRY(h, h->keys2.ensure(n));
  HIP_TRY(h, h->vals2.ensure(n));
  HIP_TRY(h, h->leaf_start.ensure((size_t)max_leaves));
  HIP_TRY(h, h->leaf_cnt.ensure((size_t)max_leaves));
  HIP_TRY(h, h->rec.ensure((size_t)max_leaves));
  HIP_TRY(h, h->cent.ensure((size_t)max_leaves * 4));
  HIP_TRY(h, h->leaf_sums.ensure((size_t)max_leaves * 9));
  HIP_TRY(h, h->run_counts.ensure((size_t)runs_blocks(n)));
  HIP_TRY(h, h->run_offsets.ensure((size_t)runs_blocks(n)));
  HIP_TRY(h, h->fin_counts.ensure((size_t)finalize_blocks(max_leaves)));
  HIP_TRY(h, h->sort_tmp.ensure(sort_temp_bytes(n)));
  // fused = launches that wait, inside the kernel, for sibling blocks (k_sort_pass, k_runs<RUNS_FUSED>)
  br.fused = fused_build_enabled();
  br.fused_sort = br.fused && fused_sort_fits(n, h->n_cus);
  // the two-launch build: steady state only (decided per attempt below), and it shares the tag table
  br.bucketed_ok = bucket_build_enabled() && bucket_build_fits(n, h->n_cus);
  // A cloud the two-launch build declined (BG_BUCKET: a bucket beyond a block's LDS or hash table, far-away coordinates)
  // is usually followed by more of its kind (the same map, the next keyframe): the attempt costs two launches and, for a
  // late decline, a full clear of the index grid, so after a decline the next 8 builds go sort-based straight away, 16 after
  // the next decline, ... at most 64.
  if (br.bucketed_ok && h->bucket_skip > 0) {
    --h->bucket_skip;
    br.bucketed_ok = false;
  }
  if ((br.fused_sort || br.bucketed_ok) && !h->sort_tags.p) {
    HIP_TRY(h, h->sort_tags.ensure(fused_table_words()));
    HIP_TRY(h, hipMemsetAsync(h->sort_tags.p, 0, h->sort_tags.cap * sizeof(uint32_t), s));
    h->sort_seq = 0;
  }
  if (br.fused && run_tag_words(n) > h->run_tags.cap) {
    HIP_TRY(h, h->run_tags.ensure(run_tag_words(n)));
    HIP_TRY(h, hipMemsetAsync(h->run_tags.p, 0, h->run_tags.cap * sizeof(uint32_t), s));
    h->run_seq = 0;
  }

  // device time of the build by HIP events when kernel timing is on (ndt_enable_kernel_timing; bench.py's
  // instrumented pass); otherwise ms_build is the wall time from the build's enqueue to its verdict and the two
  // event records, the event query and the elapsed-time call (3-4 us of host time) are saved
  const int events_tuned = tuning().build_events;
  br.build_events = events_tuned >= 0 ? events_tuned != 0 : h->timing;
  br.t0 = std::chrono::steady_clock::now();
  if (br.build_events) HIP_TRY(h, hipEventRecord(h->ev0, s));
  br.poll_done = tuning().build_wait_sync == 0 && h->prm.wait_mode == NDT_WAIT_SPIN;
  br.attempt = 0;
  return NDT_OK;
}

// The launches of one attempt.
//
// Steady state ("optimistic"): the dense grid and every scratch buffer exist from an earlier
// build, so the whole chain of launches is enqueued without waiting for the bounds -- the
// geometry and the sort plan are derived on the device (BuildGeom) and checked against the
// capacities the host assumed; ONE wait at the end (build_collect).  First build, or a cloud that
// outgrew the buffers: the host waits for the geometry once, allocates, and goes on.  A refused
// optimistic build (BG_CAPACITY / BG_PASSES) is repeated that way.
int build_enqueue(ndt_handle* h, ndt_handle::BuildRun& br) {
  hipStream_t s = h->stream;
  const float *x = br.x, *y = br.y, *z = br.z;
  const size_t n = br.n;
  const int min_pts = br.min_pts, max_leaves = br.max_leaves;
  br.optimistic = br.clean_cap != 0 && br.clean_cap == h->cell2leaf.cap;
  br.bucketed = br.bucketed_ok && br.optimistic;
  const bool optimistic = br.optimistic;
  const long long lim = std::numeric_limits<int32_t>::max();
  const long long cap_cells = optimistic ? (long long)h->cell2leaf.cap : lim;
  int passes = optimistic ? sort_passes_for_cells(cap_cells) : 0;
  h->gdh.h->status = -1;
  br.done_tag = (int)((++h->build_seq << 1) & 0x7fffffffu) | 1;  // odd: never 0, never the previous one
  const int done_tag = br.done_tag;
  h->small.h[10] = 0;
  if (br.bucketed) {
    // bounds, partition, sort, sums and statistics in two launches (k_bucket_pass, k_bucket_leaves)
    FinalizeParams fpb{h->prm.eig_inflation_ratio, h->prm.cov_mode};
    HIP_TRY(h, launch_bucket_build(x, y, z, n, br.leaf, br.inv_leaf, cap_cells, min_pts, fpb, h->gd.p, h->gdh.d, h->sort_tags.p,
                                   &h->sort_seq, h->stats.p, br.dirty_slots, h->cell2leaf.p, h->cell2leaf.cap, h->bnd.p,
                                   h->bucket_off.p, h->nleaf.p, h->tickets.p + 4, h->xyz4.p, h->leaf_sums.p, h->rec.p,
                                   h->cent.p, h->stats.p, max_leaves, h->small.d + 8, done_tag, s, h->n_cus));
  } else {
    launch_bounds_geometry(x, y, z, n, br.leaf, br.inv_leaf, cap_cells, passes, h->brows.p, h->tickets.p, h->gd.p, h->gdh.d,
                           optimistic ? h->stats.p : nullptr, optimistic ? br.dirty_slots : 0, h->cell2leaf.p,
                           h->cell2leaf.cap, h->nleaf.p, s);
    if (!optimistic) {
      HIP_TRY(h, hipStreamSynchronize(s));
      const BuildGeom& bg = *h->gdh.h;
      if (bg.status == BG_NO_FINITE) return fail(h, NDT_ERR_NO_TARGET, "target has no finite point");
      if (bg.status != BG_OK)
        return fail(h, NDT_ERR_GRID_OVERFLOW, "leaf size too small for the target extent (index overflow)");
      passes = bg.passes;
      HIP_TRY(h, h->cell2leaf.ensure((size_t)bg.g.ncells));
      HIP_TRY(h, hipMemsetAsync(h->cell2leaf.p, 0xFF, h->cell2leaf.cap * sizeof(int), s));
    }
    HIP_TRY(h, h->stats.ensure((size_t)max_leaves));
    bool in_b = false;
    if (br.fused && br.fused_sort) {
      HIP_TRY(h, sort_cloud_fused(x, y, z, n, fused_tile_for(n, h->n_cus), h->gd.p, h->gdh.d, h->xyz4.p, h->keys.p, h->keys2.p, h->vals.p, h->vals2.p,
                                  passes, h->sort_tags.p, &h->sort_seq, s, &in_b));
    } else {
      launch_cell_keys(x, y, z, n, h->gd.p, h->keys.p, h->xyz4.p, h->sort_tmp.p, s);
      HIP_TRY(h, sort_pairs(h->sort_tmp.p, h->keys.p, h->keys2.p, h->vals.p, h->vals2.p, n, passes, h->gd.p, s, &in_b));
    }
    const uint32_t* keys_sorted = in_b ? h->keys2.p : h->keys.p;
    const uint32_t* vals_sorted = in_b ? h->vals2.p : h->vals.p;
    HIP_TRY(h, launch_find_runs(keys_sorted, n, h->gd.p, h->gdh.d, min_pts, h->nleaf.p, h->run_counts.p, h->run_offsets.p,
                                h->tickets.p + 1, br.fused ? h->run_tags.p : nullptr, h->run_tags.cap, &h->run_seq,
                                h->leaf_start.p, h->leaf_cnt.p, s));
    FinalizeParams fp{h->prm.eig_inflation_ratio, h->prm.cov_mode};
    launch_finalize_leaves(h->xyz4.p, keys_sorted, vals_sorted, h->nleaf.p, h->leaf_start.p, h->leaf_cnt.p, max_leaves,
                           fp, h->leaf_sums.p, h->rec.p, h->cent.p, h->stats.p, h->cell2leaf.p, h->fin_counts.p, h->tickets.p + 2,
                           h->small.d + 8, done_tag, s);
  }  // (sort-based pipeline)
  HIP_TRY(h, hipGetLastError());
  if (br.build_events) HIP_TRY(h, hipEventRecord(h->ev1, s));
  return NDT_OK;
}

// Waits for the verdict of the attempt in flight.  0: built; 1: once more (br says how); < 0: error.
int build_collect(ndt_handle* h, ndt_handle::BuildRun& br) {
  hipStream_t s = h->stream;
  const int done_tag = br.done_tag;
  if (br.poll_done) {
    // the last block of the last kernel writes {slots, accepted, tag} to pinned memory in one
    // store: watching that word costs less than a stream synchronisation (which wakes this
    // thread through the runtime's signal); every later launch is ordered behind the build by
    // the stream anyway.  A build that does not report within 2 s is left to the runtime.
    volatile int* done = h->small.h + 10;
    const auto t_wait = std::chrono::steady_clock::now();
    unsigned spins = 0;
    while (*done != done_tag) {
      _mm_pause();
      if ((++spins & 0xfffu) == 0 &&
          std::chrono::steady_clock::now() - t_wait > std::chrono::seconds(2)) break;
    }
    std::atomic_thread_fence(std::memory_order_acquire);  // counts and geometry are read after the tag
    if (*done == done_tag) {
      if (br.build_events) {
        hipError_t q;
        while ((q = hipEventQuery(h->ev1)) == hipErrorNotReady) _mm_pause();
        HIP_TRY(h, q);
      }
    } else {
      HIP_TRY(h, hipStreamSynchronize(s));
    }
  } else {
    HIP_TRY(h, hipStreamSynchronize(s));
  }
  const BuildGeom& bg = *h->gdh.h;
  if (br.bucketed && (bg.status == BG_BUCKET || bg.status == BG_SPIN)) {
    // declined (a bucket beyond a block's LDS or table, huge coordinates) or gave up waiting for sibling
    // blocks.  Once more, sort-based.  BG_SPIN: neither launch has written a leaf and the old cells are
    // already reset; BG_BUCKET may come late (one bucket's table overflowed after others had published
    // leaves): that retry does not trust the grid -- full clear, geometry awaited.
    br.bucketed_ok = false;
    br.dirty_slots = 0;
    if (bg.status == BG_BUCKET) {
      br.clean_cap = 0;
      h->bucket_backoff = std::min(64, std::max(8, 2 * h->bucket_backoff));
      h->bucket_skip = h->bucket_backoff;
    }
    ++h->n_bucket_fallbacks;
    int rc = neutral_bounds(h);   // (the launch pair resets them itself on every path; belt and braces)
    return rc ? rc : 1;
  }
  if (br.bucketed && bg.status == BG_OK) {
    ++h->n_bucket_builds;
    h->bucket_backoff = 0;
  }
  if (br.optimistic && (bg.status == BG_CAPACITY || bg.status == BG_PASSES)) {
    // the cloud outgrew the dense grid (or the enqueued sort passes): nothing after the bounds
    // kernel ran; the old cells were reset by it.  Once more, waiting for the geometry.
    br.clean_cap = 0;
    br.dirty_slots = 0;
    return 1;
  }
  if (br.fused && bg.status == BG_SPIN) {
    // a fused launch gave up waiting for its sibling blocks (the CUs were held by other work).
    // Once more with the classic three-launch passes, which never wait inside a kernel.
    // (k_runs<RUNS_FUSED> publishes a block's tag BEFORE it waits, so its last block can set nleaf > 0
    // while a middle block gave up: k_leaf_finalize may then have written indices of stale slots into
    // the grid.  The retry therefore does not trust the grid: full clear, geometry awaited.)
    br.fused = false;
    br.dirty_slots = 0;
    br.clean_cap = 0;
    ++h->n_fused_sort_fallbacks;
    return 1;
  }
  if (bg.status == BG_NO_FINITE) return fail(h, NDT_ERR_NO_TARGET, "target has no finite point");
  if (bg.status == BG_OVERFLOW)
    return fail(h, NDT_ERR_GRID_OVERFLOW, std::string("leaf size too small for the target extent (index overflow)") +
                                              (br.bucketed ? " [two-launch build" : " [sort-based build") + ", finite points " +
                                              std::to_string(bg.n_finite) + "]");
  if (bg.status != BG_OK)
    return fail(h, NDT_ERR_HIP, "voxel-grid build ended without a verdict (internal, status " + std::to_string(bg.status) +
                                    (br.bucketed ? ", two-launch build)" : ", sort-based build)"));
  h->geom = bg.g;
  for (int a = 0; a < 3; ++a) h->max_b[a] = bg.max_b[a];
  return 0;
}

// collects the attempt in flight, repeats the build as often as its verdicts ask for, and publishes the grid
int build_complete(ndt_handle* h, ndt_handle::BuildRun& br) {
  for (;;) {
    int rc = build_collect(h, br);
    if (rc < 0) return rc;
    if (rc == 0) break;
    if (++br.attempt >= 5) return fail(h, NDT_ERR_HIP, "voxel-grid build did not go through (internal)");
    rc = build_enqueue(h, br);
    if (rc) return rc;
  }
  float ms = 0;
  if (br.build_events) HIP_TRY(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
  else ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - br.t0).count();
  h->ms_build = ms;
  h->tm.ms_last_build = ms;
  h->n_slots = h->small.h[8];
  h->grid_clean_cap = h->cell2leaf.cap;
  h->grid_dirty_slots = h->n_slots;
  h->n_valid = h->small.h[9];
  h->have_grid = true;
  h->prec_valid = false;
  if (h->record_format == NDT_RECORDS_PACKED48) {
    // behind the build on its stream; the first evaluation of an align is an ordinary launch on the same stream
    int rc = pack_records(h, false);
    if (rc) return rc;
  }
  return NDT_OK;
}

// The voxel-grid build proper; x/y/z are device pointers.  `defer`: a steady-state build is only ENQUEUED (the
// asynchronous host hand-off): the arrays must stay valid -- they are the engine's own copy there -- and the first
// call that needs the grid collects the verdict through settle().  A first build, or one that has to wait for the
// geometry anyway, completes here either way.
int build_grid(ndt_handle* h, const float* x, const float* y, const float* z, size_t n, bool defer = false) {
  h->build_pending = false;
  h->deferred_rc = 0;
  h->ms_settle_wait = 0;
  int rc = build_begin(h, x, y, z, n, h->brun);
  if (rc) return rc;
  rc = build_enqueue(h, h->brun);
  if (rc) return rc;
  if (defer && h->brun.optimistic) {
    h->build_pending = true;
    return NDT_OK;
  }
  return build_complete(h, h->brun);
}

// Completes whatever an asynchronous hand-off left in flight that the caller is about to depend on: the pending
// build's verdict (a failed build is reported HERE, by the first call that needs the grid), and the order of the
// engine's streams behind the source lane.
int settle_build(ndt_handle* h) {
  if (!h->build_pending) return NDT_OK;
  h->build_pending = false;
  const auto t0 = std::chrono::steady_clock::now();
  int rc = build_complete(h, h->brun);
  h->ms_settle_wait = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  if (rc) {  // kept for the first call that needs the grid (the collecting call may be a setter that does not)
    h->deferred_rc = rc;
    h->deferred_msg = h->err;
  }
  return rc;
}

int settle_source(ndt_handle* h) {
  if (!h->src_upload_pending) return NDT_OK;
  h->src_upload_pending = false;
  HIP_TRY(h, hipStreamWaitEvent(h->stream, h->lane_s.done, 0));
  HIP_TRY(h, hipStreamWaitEvent(h->stream2, h->lane_s.done, 0));
  return NDT_OK;
}

// The source lane's stream starts behind the target lane's TRANSFER (not its build): both transfers share one PCIe
// link, the target's is on the critical path (the build waits for it; the source is not needed before the first
// evaluation), and the source's 2.4 MB then cross while the build runs and the link is idle.
int source_behind_target_transfer(ndt_handle* h, bool async) {
  if (async && h->lane_t.busy) HIP_TRY(h, hipStreamWaitEvent(h->ustream, h->lane_t.done, 0));
  return NDT_OK;
}

// A deferred build's failure is reported ONCE, by the first call that needs the grid (or by ndt_wait); from then on the
// handle is where a failed blocking ndt_set_target leaves it: no grid (NDT_ERR_NO_TARGET for consumers), nothing pending.
int report_deferred(ndt_handle* h) {
  const int rc = h->deferred_rc;
  const std::string msg = h->deferred_msg;
  h->deferred_rc = 0;
  h->deferred_msg.clear();
  return fail(h, rc, msg);
}

// for calls that need the grid: a deferred build's failure is theirs to report
int settle(ndt_handle* h) {
  (void)settle_build(h);
  int rs = settle_source(h);
  if (!h->have_grid && h->deferred_rc) return report_deferred(h);
  return rs;
}

// for calls that borrow the build's scratch but leave the target alone: the pending build is completed and, if it
// failed, its verdict kept for the first call that needs the grid
void settle_discard_keep_grid(ndt_handle* h) { (void)settle_build(h); }

// for calls that replace the target: the pending build is completed (its scratch and the engine's own copy of
// the cloud are about to be reused) and its verdict dropped
void settle_discard(ndt_handle* h) {
  (void)settle_build(h);
  h->deferred_rc = 0;
  h->deferred_msg.clear();
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
static bool source_sort_wanted(const ndt_handle* h, int n_valid) {
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
static bool first_eval_behind_build(const ndt_handle* h) {
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

int wait_slots(ndt_handle* h, unsigned long long seq, int K = 1, int first = 0) {
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
int evaluate(ndt_handle* h, const double p[6], const float T[16], bool need_h, Eval* out, bool score_only = false,
             bool safe_retry = false) {
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
  // never produces a non-finite score from finite records
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

}  // namespace

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
extern "C" {

int ndt_abi_version(void) { return NDT_HIP_ABI_VERSION; }

void ndt_default_params(ndt_params* p) {
  if (!p) return;
  std::memset(p, 0, sizeof(*p));
  p->resolution = 1.0f;
  p->step_size = 0.1;
  p->trans_epsilon = 0.1;   // pclomp's constructor default
  p->max_iterations = 35;
  p->outlier_ratio = 0.55;
  p->search_method = NDT_DIRECT7;
  p->min_points_per_voxel = 6;
  p->eig_inflation_ratio = 0.01;
  p->hessian_mode = NDT_HESSIAN_FULL;
  p->cov_mode = NDT_COV_SVN;
  p->add_ridge = 0;
  p->use_line_search = 1;
  p->regularization_scale_factor = 0.0f;
  p->num_threads = 1;
  p->device_id = -1;
  p->wait_mode = NDT_WAIT_SPIN;
  p->source_order = NDT_SOURCE_ORDER_AUTO;
  p->prelaunch = NDT_PRELAUNCH_AUTO;
}

int ndt_params_preset(ndt_params* p, int preset) {
  if (!p) return NDT_ERR_INVALID_ARG;
  switch (preset) {
    case NDT_PRESET_DEFAULT:
      p->cov_mode = NDT_COV_SVN; p->hessian_mode = NDT_HESSIAN_FULL; p->add_ridge = 0; p->use_line_search = 1;
      p->min_points_per_voxel = 6;
      return NDT_OK;
    case NDT_PRESET_PCLOMP_RECALLED:
      p->cov_mode = NDT_COV_PCL_RECALLED; p->hessian_mode = NDT_HESSIAN_FULL; p->add_ridge = 0; p->use_line_search = 1;
      p->min_points_per_voxel = 6;
      return NDT_OK;
    case NDT_PRESET_SVN:  // ref: svn_ndt.h:314, svn_ndt_impl.hpp:650-653, voxel_grid_covariance_impl.hpp:287-291
      p->cov_mode = NDT_COV_SVN; p->hessian_mode = NDT_HESSIAN_GAUSS_NEWTON; p->add_ridge = 1; p->use_line_search = 1;
      p->min_points_per_voxel = 6;
      return NDT_OK;
  }
  return NDT_ERR_INVALID_ARG;
}

int ndt_backend_info(char* buf, size_t cap) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    if (buf && cap) snprintf(buf, cap, "no HIP device (%s)", e == hipSuccess ? "count 0" : hipGetErrorString(e));
    return e == hipSuccess ? 0 : NDT_ERR_NO_DEVICE;
  }
  hipDeviceProp_t prop;
  if (buf && cap) {
    if (hipGetDeviceProperties(&prop, 0) == hipSuccess)
      snprintf(buf, cap, "%d device(s); device 0: %s (%s), %d CUs, %.1f GiB", n, prop.name,
               prop.gcnArchName, prop.multiProcessorCount, (double)prop.totalGlobalMem / (1 << 30));
    else
      snprintf(buf, cap, "%d device(s)", n);
  }
  return n;
}

int ndt_create(const ndt_params* p, ndt_handle** out) {
  if (!out) return NDT_ERR_INVALID_ARG;
  *out = nullptr;
  ndt_params prm;
  if (p) prm = *p; else ndt_default_params(&prm);
  std::string why;
  if (!params_valid(&prm, &why)) return NDT_ERR_INVALID_ARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return NDT_ERR_NO_DEVICE;
  int dev = prm.device_id;
  if (dev < 0 && hipGetDevice(&dev) != hipSuccess) return NDT_ERR_NO_DEVICE;
  if (dev >= ndev) return NDT_ERR_INVALID_ARG;
  if (hipSetDevice(dev) != hipSuccess) return NDT_ERR_NO_DEVICE;
  ndt_handle* h = new ndt_handle();
  h->prm = prm;
  h->device = dev;
  {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) h->n_cus = cus;
  }
  {  // A/B switches of the pose hand-over to pre-launched kernels (ndt_tuning; profiles/r02_mailbox_ab.txt)
    const ndt_tuning& tn = tuning();
    h->mbox_tagged = tn.mbox_tagged != 0;
    h->mbox_preload = tn.mbox_preload != 0;
    h->two_streams = tn.prelaunch_streams != 1;
  }
  {
    const char* e = getenv("NDT_HANDOFF");  // operational knob: "sync" = the blocking hand-off of rounds 1-3
    if (e && std::strcmp(e, "sync") == 0) h->handoff_mode = NDT_HANDOFF_SYNC;
  }
  if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess ||
      hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking) != hipSuccess ||
      hipStreamCreateWithFlags(&h->ustream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreate(&h->ev0) != hipSuccess || hipEventCreate(&h->ev1) != hipSuccess ||
      hipEventCreate(&h->ev2) != hipSuccess) {
    delete h;
    return NDT_ERR_HIP;
  }
  *out = h;
  return NDT_OK;
}

int ndt_destroy(ndt_handle* h) {
  if (!h) return NDT_OK;
  h->keepwarm.stop();
  (void)hipSetDevice(h->device);
  settle_discard(h);
  if (h->ustream) (void)hipStreamSynchronize(h->ustream);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  if (h->stream2) (void)hipStreamSynchronize(h->stream2);
  h->pool.reset();
  for (ndt_handle::UploadLane* lane : {&h->lane_t, &h->lane_s}) {
    lane->stage.release();
    if (lane->done) (void)hipEventDestroy(lane->done);
    if (lane->t0) (void)hipEventDestroy(lane->t0);
    if (lane->t1) (void)hipEventDestroy(lane->t1);
  }
  h->red.destroy();
  h->tx.release(); h->ty.release(); h->tz.release();
  h->keys.release(); h->vals.release(); h->keys2.release(); h->vals2.release();
  h->sort_tmp.release(); h->nleaf.release(); h->leaf_start.release(); h->leaf_cnt.release();
  h->cell2leaf.release(); h->rec.release(); h->prec.release(); h->prec_valid = false; h->cent.release(); h->stats.release();
  h->run_counts.release(); h->run_offsets.release(); h->fin_counts.release(); h->bucket_off.release(); h->bnd.release(); h->sort_tags.release(); h->run_tags.release(); h->xyz4.release(); h->leaf_sums.release();
  h->brows.release(); h->tickets.release(); h->gd.release(); h->gdh.release();
  for (auto& kv : h->keyframes) { kv.second.x.release(); kv.second.y.release(); kv.second.z.release(); }
  h->keyframes.clear();
  for (auto& kf : h->keyframe_pool) { kf.x.release(); kf.y.release(); kf.z.release(); }
  h->keyframe_pool.clear();
  h->sx.release(); h->sy.release(); h->sz.release();
  h->ox.release(); h->oy.release(); h->oz.release(); h->skeys.release(); h->skeys2.release();
  h->svals.release(); h->svals2.release(); h->ssort_tmp.release(); h->splan.release();
  h->result.release(); h->small.release(); h->partials.release();
  h->dres.release(); h->dposes.release(); h->hposes.release(); h->counters.release(); h->flag.release();
  if (h->mbox) (void)hipFree(h->mbox);
  if (h->bposes) (void)hipFree(h->bposes);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  if (h->ev2) (void)hipEventDestroy(h->ev2);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  if (h->stream2) (void)hipStreamDestroy(h->stream2);
  if (h->ustream) (void)hipStreamDestroy(h->ustream);
  h->arrive_ctr.release(); h->arrived.release();
  delete h;
  return NDT_OK;
}

const char* ndt_last_error(const ndt_handle* h) { return h ? h->err.c_str() : "null handle"; }

int ndt_get_params(const ndt_handle* h, ndt_params* p) {
  if (!h || !p) return NDT_ERR_INVALID_ARG;
  *p = h->prm;
  return NDT_OK;
}

int ndt_set_params(ndt_handle* h, const ndt_params* p) {
  if (!h || !p) return NDT_ERR_INVALID_ARG;
  std::string why;
  if (!params_valid(p, &why)) return fail(h, NDT_ERR_INVALID_ARG, why);
  if (h->build_pending) {
    int rc0 = bind_device(h);
    if (rc0) return rc0;
    (void)settle_build(h);  // (a failure stays with the handle for the first call that needs the grid)
  }
  const bool grid_changed = h->have_grid && (std::fabs(p->resolution - h->prm.resolution) > 1e-6f ||
                                             p->min_points_per_voxel != h->prm.min_points_per_voxel ||
                                             p->eig_inflation_ratio != h->prm.eig_inflation_ratio ||
                                             p->cov_mode != h->prm.cov_mode);
  const bool rebuild = grid_changed && h->tx.p && h->n_tgt > 0;
  const int dev = h->prm.device_id;
  if (p->source_order != h->prm.source_order) h->src_sorted = false;
  h->prm = *p;
  h->prm.device_id = dev;  // a handle never migrates
  h->prelaunch_suspended = false;  // the caller has spoken: try again
  h->prelaunch_strikes = 0;
  if (grid_changed && !rebuild) {
    // the target came through ndt_set_target_device and was consumed there: the grid cannot be
    // re-voxelised, and the old one must not be evaluated with the new constants.  The next
    // align / eval reports NDT_ERR_NO_TARGET until a target is set again.
    h->have_grid = false;
    h->n_valid = 0;
    return NDT_OK;
  }
  if (rebuild) {
    // setResolution on a loaded target re-voxelises it (ref: svn_ndt_impl.hpp:162-176)
    int rc = bind_device(h);
    if (rc) return rc;
    return build_grid(h, h->tx.p, h->ty.p, h->tz.p, h->n_tgt);
  }
  return NDT_OK;
}

int ndt_set_target(ndt_handle* h, const float* xyz, size_t n, size_t stride_bytes) {
  if (!h || (!xyz && n) || stride_bytes < 12 || stride_bytes % 4) return NDT_ERR_INVALID_ARG;
  int rc = bind_device(h);
  if (rc) return rc;
  settle_discard(h);
  const bool async = h->handoff_mode == NDT_HANDOFF_ASYNC;
  rc = upload_soa(h, h->lane_t, h->stream, xyz, nullptr, nullptr, nullptr, n, stride_bytes, h->tx, h->ty, h->tz, !async);
  if (rc) return rc;
  return build_grid(h, h->tx.p, h->ty.p, h->tz.p, n, async);
}

int ndt_set_target_soa(ndt_handle* h, const float* x, const float* y, const float* z, size_t n) {
  if (!h || ((!x || !y || !z) && n)) return NDT_ERR_INVALID_ARG;
  int rc = bind_device(h);
  if (rc) return rc;
  settle_discard(h);
  const bool async = h->handoff_mode == NDT_HANDOFF_ASYNC;
  rc = upload_soa(h, h->lane_t, h->stream, nullptr, x, y, z, n, 0, h->tx, h->ty, h->tz, !async);
  if (rc) return rc;
  return build_grid(h, h->tx.p, h->ty.p, h->tz.p, n, async);
}

int ndt_set_target_device(ndt_handle* h, const float* dx, const float* dy, const float* dz, size_t n) {
  if (!h || ((!dx || !dy || !dz) && n)) return NDT_ERR_INVALID_ARG;
  int rc = bind_device(h);
  if (rc) return rc;
  // built straight from the caller's arrays; nothing is retained, so a later
  // resolution change cannot re-voxelise this target
  settle_discard(h);
  h->tx.release(); h->ty.release(); h->tz.release();
  return build_grid(h, dx, dy, dz, n);
}

int ndt_set_target_device_deferred(ndt_handle* h, const float* dx, const float* dy, const float* dz, size_t n) {
  if (!h || ((!dx || !dy || !dz) && n)) return NDT_ERR_INVALID_ARG;
  int rc = bind_device(h);
  if (rc) return rc;
  settle_discard(h);
  h->tx.release(); h->ty.release(); h->tz.release();
  // (a first build, or one that has to wait for the geometry anyway, completes inside the call: build_grid)
  return build_grid(h, dx, dy, dz, n, h->handoff_mode == NDT_HANDOFF_ASYNC);
}

int ndt_set_source(ndt_handle* h, const float* xyz, size_t n, size_t stride_bytes) {
  if (!h || (!xyz && n) || stride_bytes < 12 || stride_bytes % 4) return NDT_ERR_INVALID_ARG;
  int rc = bind_device(h);
  if (rc) return rc;
  h->vx = h->vy = h->vz = nullptr;
  h->n_src = 0;
  // asynchronous hand-off: on the source lane's own stream, so that its copies run beside the target's build; the
  // engine's streams are ordered behind it by the first call that evaluates (settle_source)
  const bool async = h->handoff_mode == NDT_HANDOFF_ASYNC;
  rc = source_behind_target_transfer(h, async);
  if (rc) return rc;
  rc = upload_soa(h, h->lane_s, async ? h->ustream : h->stream, xyz, nullptr, nullptr, nullptr, n, stride_bytes, h->sx, h->sy, h->sz, !async);
  if (rc) return rc;
  h->src_upload_pending = async && n > 0;
  h->vx = h->sx.p; h->vy = h->sy.p; h->vz = h->sz.p;
  h->n_src = n;
  h->src_sorted = false;
  return NDT_OK;
}

int ndt_set_source_soa(ndt_handle* h, const float* x, const float* y, const float* z, size_t n) {
  if (!h || ((!x || !y || !z) && n)) return NDT_ERR_INVALID_ARG;
  int rc = bind_device(h);
  if (rc) return rc;
  h->vx = h->vy = h->vz = nullptr;
  h->n_src = 0;
  const bool async = h->handoff_mode == NDT_HANDOFF_ASYNC;
  rc = source_behind_target_transfer(h, async);
  if (rc) return rc;
  rc = upload_soa(h, h->lane_s, async ? h->ustream : h->stream, nullptr, x, y, z, n, 0, h->sx, h->sy, h->sz, !async);
  if (rc) return rc;
  h->src_upload_pending = async && n > 0;
  h->vx = h->sx.p; h->vy = h->sy.p; h->vz = h->sz.p;
  h->n_src = n;
  h->src_sorted = false;
  return NDT_OK;
}

int ndt_set_source_device(ndt_handle* h, const float* dx, const float* dy, const float* dz, size_t n) {
  if (!h || ((!dx || !dy || !dz) && n)) return NDT_ERR_INVALID_ARG;
  int rc = bind_device(h);
  if (rc) return rc;
  rc = settle_source(h);  // a source hand-off still in flight writes the same arrays on its own stream
  if (rc) return rc;
  HIP_TRY(h, h->sx.ensure(n));
  HIP_TRY(h, h->sy.ensure(n));
  HIP_TRY(h, h->sz.ensure(n));
  if (n) {
    launch_copy_soa(dx, dy, dz, n, h->sx.p, h->sy.p, h->sz.p, h->stream);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipStreamSynchronize(h->stream));  // the caller's arrays are consumed during the call
  }
  h->vx = h->sx.p; h->vy = h->sy.p; h->vz = h->sz.p;
  h->n_src = n;
  h->src_sorted = false;
  return NDT_OK;
}

// pclomp's setInputSource keeps the caller's shared_ptr, not a copy (ref: pcl::Registration::
// setInputSource, called at run/pipeline.cpp:558): the same contract for device-resident arrays.
int ndt_set_source_device_view(ndt_handle* h, const float* dx, const float* dy, const float* dz, size_t n) {
  if (!h || ((!dx || !dy || !dz) && n)) return NDT_ERR_INVALID_ARG;
  int rc = bind_device(h);
  if (rc) return rc;
  h->vx = dx; h->vy = dy; h->vz = dz;
  h->n_src = n;
  h->src_sorted = false;
  return NDT_OK;
}

// The caller has rewritten the arrays of a viewed source in place (a reused scan buffer): the cached
// block-ordered copy (maybe_sort_source) is stale.  Cheap: no copy, no launch.
int ndt_source_changed(ndt_handle* h) {
  if (!h) return NDT_ERR_INVALID_ARG;
  h->src_sorted = false;
  return NDT_OK;
}

// 48-byte packed voxel records (f64 mean, f32 inverse covariance) instead of the 80-byte f64 ones: three 16-byte
// loads per neighbour instead of five.  Takes effect at the next evaluation; the statistics the engine exports
// (ndt_get_leaves) are the f64 ones either way.
int ndt_set_record_format(ndt_handle* h, int format) {
  if (!h) return NDT_ERR_INVALID_ARG;
  if (format != NDT_RECORDS_F64 && format != NDT_RECORDS_PACKED48) return fail(h, NDT_ERR_INVALID_ARG, "unknown record format");
  h->record_format = format;
  return NDT_OK;
}

int ndt_get_record_format(const ndt_handle* h) { return h ? h->record_format : NDT_ERR_INVALID_ARG; }

// ---- multi-grid target [RECALLED: tier4 ndt_omp multigrid_ndt_omp.h / multi_voxel_grid_covariance_omp.h,
// named by the reference's build (CMakeLists.txt:41-42), sources in the absent submodule] ----------
// addTarget(cloud, id): the cloud is voxelised ON ITS OWN by the ordinary build (same kernels, same
// leaf statistics as setInputTarget) and its valid leaves are kept under `id`.
int ndt_multigrid_add_target(ndt_handle* h, int64_t id, const float* xyz, size_t n, size_t stride_bytes) {
  if (!h || !xyz || n == 0 || stride_bytes < 12 || stride_bytes % 4) return NDT_ERR_INVALID_ARG;
  int rc = bind_device(h);
  if (rc) return rc;
  settle_discard(h);
  rc = upload_soa(h, h->lane_t, h->stream, xyz, nullptr, nullptr, nullptr, n, stride_bytes, h->tx, h->ty, h->tz, true);
  if (rc) return rc;
  rc = build_grid(h, h->tx.p, h->ty.p, h->tz.p, n);
  // the handle's single-grid table is scratch here: whatever happens, it is not a target to align to,
  // and the cloud is not kept (a later resolution change cannot silently re-voxelise one tile)
  h->have_grid = false;
  h->tx.release(); h->ty.release(); h->tz.release();
  if (rc) return rc;
  std::vector<LeafStats> st((size_t)h->n_slots);
  std::vector<VoxelRecord> rec((size_t)h->n_slots);
  if (h->n_slots) {
    HIP_TRY(h, hipMemcpy(st.data(), h->stats.p, st.size() * sizeof(LeafStats), hipMemcpyDeviceToHost));
    HIP_TRY(h, hipMemcpy(rec.data(), h->rec.p, rec.size() * sizeof(VoxelRecord), hipMemcpyDeviceToHost));
  }
  ndt_handle::MultiGridEntry e;
  const GridGeom& g = h->geom;
  for (size_t s = 0; s < st.size(); ++s) {
    if (st[s].count <= 0) continue;
    const int c = st[s].cell;
    e.ijk.push_back(g.min_b[0] + c % g.div_b[0]);
    e.ijk.push_back(g.min_b[1] + (c / g.div_b[0]) % g.div_b[1]);
    e.ijk.push_back(g.min_b[2] + c / g.mul2);
    e.rec.push_back(rec[s]);
    e.stats.push_back(st[s]);
  }
  e.n_points = n;
  e.resolution = h->prm.resolution;
  e.min_points = h->prm.min_points_per_voxel;
  e.cov_mode = h->prm.cov_mode;
  e.eig_ratio = h->prm.eig_inflation_ratio;
  h->mgrids[id] = std::move(e);
  h->multi_active = false;  // the union has to be re-assembled (createVoxelKdtree)
  return NDT_OK;
}

int ndt_multigrid_remove_target(ndt_handle* h, int64_t id) {
  if (!h) return NDT_ERR_INVALID_ARG;
  auto it = h->mgrids.find(id);
  if (it == h->mgrids.end()) return fail(h, NDT_ERR_INVALID_ARG, "unknown multi-grid target id");
  h->mgrids.erase(it);
  if (h->multi_active) { h->multi_active = false; h->have_grid = false; }
  return NDT_OK;
}

int64_t ndt_multigrid_count(const ndt_handle* h) { return h ? (int64_t)h->mgrids.size() : NDT_ERR_INVALID_ARG; }

// createVoxelKdtree(): the union of all stored grids becomes the device table.  All grids sit on the
// same absolute lattice (voxel = floor(p / leaf), ref: voxel_grid_covariance_impl.hpp:222-225), so the
// kd-tree over every grid's centroids is again a 27-cell scan + distance test; a cell holds one leaf
// per grid that has points there (head in the dense index, the others chained through the records).
int ndt_multigrid_create_kdtree(ndt_handle* h) {
  if (!h) return NDT_ERR_INVALID_ARG;
  int rc = bind_device(h);
  if (rc) return rc;
  settle_discard(h);
  h->have_grid = false;
  h->multi_active = false;
  h->src_sorted = false;
  if (h->mgrids.empty()) return fail(h, NDT_ERR_NO_TARGET, "no multi-grid target (addTarget first)");
  const auto t_begin = std::chrono::steady_clock::now();
  long long mn[3] = {LLONG_MAX, LLONG_MAX, LLONG_MAX}, mx[3] = {LLONG_MIN, LLONG_MIN, LLONG_MIN};
  size_t total = 0, total_points = 0;
  for (const auto& kv : h->mgrids) {
    const auto& e = kv.second;
    if (e.resolution != h->prm.resolution || e.min_points != h->prm.min_points_per_voxel || e.cov_mode != h->prm.cov_mode ||
        e.eig_ratio != h->prm.eig_inflation_ratio)
      return fail(h, NDT_ERR_INVALID_ARG, "a multi-grid target was voxelised with other grid parameters: add it again");
    for (size_t i = 0; i < e.rec.size(); ++i)
      for (int a = 0; a < 3; ++a) {
        mn[a] = std::min<long long>(mn[a], e.ijk[3 * i + a]);
        mx[a] = std::max<long long>(mx[a], e.ijk[3 * i + a]);
      }
    total += e.rec.size();
    total_points += e.n_points;
  }
  if (total == 0) return fail(h, NDT_ERR_NO_TARGET, "the multi-grid targets hold no valid voxel");
  long long d[3], ncells = 1;
  for (int a = 0; a < 3; ++a) { d[a] = mx[a] - mn[a] + 1; ncells *= d[a]; if (ncells >= 2147483647ll) break; }
  if (ncells >= 2147483647ll || total >= (size_t)2147483647)
    return fail(h, NDT_ERR_GRID_OVERFLOW, "the multi-grid targets span too many cells (index overflow)");
  GridGeom g{};
  g.leaf = h->prm.resolution;
  g.inv_leaf = 1.0f / h->prm.resolution;
  for (int a = 0; a < 3; ++a) {
    g.min_b[a] = (int)mn[a];
    g.div_b[a] = (int)d[a];
    g.lo[a] = (float)g.min_b[a] * g.leaf;
    g.hi[a] = (float)(mx[a] + 1) * g.leaf;
  }
  g.mul1 = g.div_b[0];
  g.mul2 = g.div_b[0] * g.div_b[1];
  g.ncells = (int)ncells;
  // (cell, grid, leaf) order: grids in ascending id, leaves in their own ascending cell order
  struct Ref { int cell; const ndt_handle::MultiGridEntry* e; size_t i; };
  std::vector<Ref> refs;
  refs.reserve(total);
  for (const auto& kv : h->mgrids) {
    const auto& e = kv.second;
    for (size_t i = 0; i < e.rec.size(); ++i) {
      const int c = (e.ijk[3 * i] - g.min_b[0]) + (e.ijk[3 * i + 1] - g.min_b[1]) * g.mul1 + (e.ijk[3 * i + 2] - g.min_b[2]) * g.mul2;
      refs.push_back(Ref{c, &e, i});
    }
  }
  std::stable_sort(refs.begin(), refs.end(), [](const Ref& a, const Ref& b) { return a.cell < b.cell; });
  std::vector<VoxelRecord> rec(total);
  std::vector<int> head_cells, head_slots;
  h->multi_stats.resize(total);
  for (size_t s = 0; s < total; ++s) {
    rec[s] = refs[s].e->rec[refs[s].i];
    rec[s].pad = (s + 1 < total && refs[s + 1].cell == refs[s].cell) ? (double)(s + 1) : -1.0;
    h->multi_stats[s] = refs[s].e->stats[refs[s].i];
    h->multi_stats[s].cell = refs[s].cell;
    if (s == 0 || refs[s - 1].cell != refs[s].cell) { head_cells.push_back(refs[s].cell); head_slots.push_back((int)s); }
  }
  hipStream_t s = h->stream;
  HIP_TRY(h, h->cell2leaf.ensure((size_t)g.ncells));
  HIP_TRY(h, hipMemsetAsync(h->cell2leaf.p, 0xFF, h->cell2leaf.cap * sizeof(int), s));
  h->grid_clean_cap = 0;   // the next ordinary build starts from a cleared index
  h->grid_dirty_slots = 0;
  HIP_TRY(h, h->rec.ensure(total));
  HIP_TRY(h, hipMemcpyAsync(h->rec.p, rec.data(), total * sizeof(VoxelRecord), hipMemcpyHostToDevice, s));
  // f32 centroids + the chain link of every leaf (the slot of the next leaf in the same cell, -1: none), as bits
  std::vector<float> cent(4 * total);
  for (size_t k = 0; k < total; ++k) {
    cent[4 * k + 0] = (float)rec[k].mean[0];
    cent[4 * k + 1] = (float)rec[k].mean[1];
    cent[4 * k + 2] = (float)rec[k].mean[2];
    const int link = (int)rec[k].pad;
    std::memcpy(&cent[4 * k + 3], &link, sizeof(int));
  }
  HIP_TRY(h, h->cent.ensure(4 * total));
  HIP_TRY(h, hipMemcpyAsync(h->cent.p, cent.data(), cent.size() * sizeof(float), hipMemcpyHostToDevice, s));
  DevBuf<int> dc, ds;
  HIP_TRY(h, dc.ensure(head_cells.size()));
  HIP_TRY(h, ds.ensure(head_cells.size()));
  hipError_t e1 = hipMemcpyAsync(dc.p, head_cells.data(), head_cells.size() * sizeof(int), hipMemcpyHostToDevice, s);
  if (e1 == hipSuccess) e1 = hipMemcpyAsync(ds.p, head_slots.data(), head_slots.size() * sizeof(int), hipMemcpyHostToDevice, s);
  if (e1 == hipSuccess) { launch_scatter_heads(dc.p, ds.p, head_cells.size(), h->cell2leaf.p, s); e1 = hipGetLastError(); }
  if (e1 == hipSuccess) e1 = hipStreamSynchronize(s);
  dc.release(); ds.release();
  HIP_TRY(h, e1);
  h->geom = g;
  for (int a = 0; a < 3; ++a) h->max_b[a] = (int)mx[a];
  h->n_slots = h->n_valid = (int)total;
  h->n_tgt = total_points;
  h->ms_build = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
  h->tm.ms_last_build = h->ms_build;
  h->have_grid = true;
  h->multi_active = true;
  h->prec_valid = false;
  return NDT_OK;
}

int ndt_keyframe_put(ndt_handle* h, int64_t id, const float* xyz, size_t n, size_t stride_bytes) {
  if (!h || (!xyz && n) || stride_bytes < 12 || stride_bytes % 4) return NDT_ERR_INVALID_ARG;
  int rc = bind_device(h);
  if (rc) return rc;
  const bool fresh = h->keyframes.find(id) == h->keyframes.end();
  ndt_handle::Keyframe& kf = h->keyframes[id];
  if (fresh) {   // the buffers of an erased keyframe that are large enough, if any (the stream orders their reuse)
    for (size_t i = 0; i < h->keyframe_pool.size(); ++i)
      if (h->keyframe_pool[i].x.cap >= n && h->keyframe_pool[i].y.cap >= n && h->keyframe_pool[i].z.cap >= n) {
        kf = h->keyframe_pool[i];
        h->keyframe_pool.erase(h->keyframe_pool.begin() + (long)i);
        break;
      }
  }
  if (kf.x.p && h->vx == kf.x.p) {   // the keyframe being replaced is the viewed source: it has to be set again
    h->vx = h->vy = h->vz = nullptr;
    h->n_src = 0;
  }
  // like a host hand-off: the caller's cloud is consumed when the call returns, the transfer runs behind it on the
  // engine's stream, where everything that reads the archive is enqueued too
  rc = upload_soa(h, h->lane_t, h->stream, xyz, nullptr, nullptr, nullptr, n, stride_bytes, kf.x, kf.y, kf.z,
                  h->handoff_mode != NDT_HANDOFF_ASYNC);
  if (rc) return rc;
  kf.n = n;
  return NDT_OK;
}

int ndt_set_source_from_keyframe(ndt_handle* h, int64_t id) {
  if (!h) return NDT_ERR_INVALID_ARG;
  auto it = h->keyframes.find(id);
  if (it == h->keyframes.end()) return fail(h, NDT_ERR_INVALID_ARG, "unknown keyframe id");
  const ndt_handle::Keyframe& kf = it->second;
  // a VIEW of the archived scan, not a copy (it stays the source until the keyframe is erased or replaced, which
  // unsets it): the archive is the engine's own memory and is written on the stream the evaluations run on
  return ndt_set_source_device_view(h, kf.x.p, kf.y.p, kf.z.p, kf.n);
}

int ndt_keyframe_erase(ndt_handle* h, int64_t id) {
  if (!h) return NDT_ERR_INVALID_ARG;
  auto it = h->keyframes.find(id);
  if (it == h->keyframes.end()) return NDT_ERR_INVALID_ARG;
  (void)hipSetDevice(h->device);
  if (it->second.x.p && h->vx == it->second.x.p) {   // the viewed source goes with its keyframe
    h->vx = h->vy = h->vz = nullptr;
    h->n_src = 0;
  }
  if (h->keyframe_pool.size() < 4 && it->second.x.p) {
    // (whatever still reads these arrays was enqueued on the engine's stream before this call; the next put writes
    // them on the same stream, behind it)
    it->second.n = 0;
    h->keyframe_pool.push_back(it->second);
  } else {
    it->second.x.release(); it->second.y.release(); it->second.z.release();
  }
  h->keyframes.erase(it);
  return NDT_OK;
}

int64_t ndt_keyframe_count(const ndt_handle* h) { return h ? (int64_t)h->keyframes.size() : NDT_ERR_INVALID_ARG; }

int ndt_set_target_from_keyframes(ndt_handle* h, const int64_t* ids, const double* poses16, int n_keyframes) {
  if (!h || !ids || !poses16 || n_keyframes <= 0) return NDT_ERR_INVALID_ARG;
  int rc = bind_device(h);
  if (rc) return rc;
  size_t total = 0;
  for (int k = 0; k < n_keyframes; ++k) {
    auto it = h->keyframes.find(ids[k]);
    if (it == h->keyframes.end()) return fail(h, NDT_ERR_INVALID_ARG, "unknown keyframe id");
    total += it->second.n;
  }
  settle_discard(h);
  HIP_TRY(h, h->tx.ensure(total));
  HIP_TRY(h, h->ty.ensure(total));
  HIP_TRY(h, h->tz.ensure(total));
  size_t off = 0;
  for (int k = 0; k < n_keyframes; ++k) {  // appended in the caller's order, like `target += cloud`
    const ndt_handle::Keyframe& kf = h->keyframes[ids[k]];
    launch_transform_append(kf.x.p, kf.y.p, kf.z.p, kf.n, poses16 + 16 * (size_t)k, h->tx.p + off, h->ty.p + off,
                            h->tz.p + off, h->stream);
    off += kf.n;
  }
  HIP_TRY(h, hipGetLastError());
  // (the assembled cloud is the engine's own: the build may stay in flight like a host hand-off's)
  return build_grid(h, h->tx.p, h->ty.p, h->tz.p, total, h->handoff_mode == NDT_HANDOFF_ASYNC);
}

// pcl::VoxelGrid on the device (ref: run/pipeline_ins_map_distribution.cpp:324-340, leaf = mapvoxelsize): bounds ->
// keys -> stable radix sort -> runs -> centroids with the build's own launch-per-phase kernels (min_pts = 1, no
// statistics).  Uses the build's scratch; the handle's target grid and its source are left as they are.
static int voxel_downsample_device_impl(ndt_handle* h, const float* dx, const float* dy, const float* dz, const float* di,
                                        size_t n, float leaf, float* ox, float* oy, float* oz, float* oi, size_t cap,
                                        size_t* n_out) {
  *n_out = 0;
  if (n == 0) return NDT_OK;
  if (n > (size_t)std::numeric_limits<int>::max() / 2) return fail(h, NDT_ERR_INVALID_ARG, "cloud too large");
  settle_discard_keep_grid(h);
  hipStream_t s = h->stream;
  HIP_TRY(h, h->brows.ensure(8 * (size_t)std::max(bounds_rows(n), bucket_build_tiles(n))));
  HIP_TRY(h, h->gd.ensure(1));
  HIP_TRY(h, h->gdh.ensure(1));
  if (!h->tickets.p) {
    HIP_TRY(h, h->tickets.ensure(6));
    HIP_TRY(h, hipMemsetAsync(h->tickets.p, 0, h->tickets.cap * sizeof(unsigned int), s));
  }
  HIP_TRY(h, h->nleaf.ensure(4));
  HIP_TRY(h, h->keys.ensure(n));
  HIP_TRY(h, h->xyz4.ensure(4 * n));
  HIP_TRY(h, h->vals.ensure(n));
  HIP_TRY(h, h->keys2.ensure(n));
  HIP_TRY(h, h->vals2.ensure(n));
  HIP_TRY(h, h->leaf_start.ensure(n + 1));
  HIP_TRY(h, h->leaf_cnt.ensure(n + 1));
  HIP_TRY(h, h->run_counts.ensure((size_t)runs_blocks(n)));
  HIP_TRY(h, h->run_offsets.ensure((size_t)runs_blocks(n)));
  HIP_TRY(h, h->sort_tmp.ensure(sort_temp_bytes(n)));
  HIP_TRY(h, h->small.ensure(16));
  h->gdh.h->status = -1;
  // the geometry is awaited (its pass count sizes the sort): a full, launch-per-phase pipeline that never waits
  // inside a kernel; no cell of the handle's index grid is touched (old_stats = null, no dirty slots)
  launch_bounds_geometry(dx, dy, dz, n, leaf, 1.0f / leaf, (long long)std::numeric_limits<int32_t>::max(), 0, h->brows.p,
                         h->tickets.p, h->gd.p, h->gdh.d, nullptr, 0, nullptr, 0, h->nleaf.p, s);
  HIP_TRY(h, hipGetLastError());
  HIP_TRY(h, hipStreamSynchronize(s));
  const BuildGeom bg = *h->gdh.h;
  if (bg.status == BG_NO_FINITE) return NDT_OK;   // nothing finite: an empty output, as PCL's filter leaves it
  if (bg.status != BG_OK)
    return fail(h, NDT_ERR_GRID_OVERFLOW, "leaf size too small for the cloud's extent (index overflow; PCL's VoxelGrid refuses the same cloud)");
  launch_cell_keys(dx, dy, dz, n, h->gd.p, h->keys.p, h->xyz4.p, h->sort_tmp.p, s);
  bool in_b = false;
  HIP_TRY(h, sort_pairs(h->sort_tmp.p, h->keys.p, h->keys2.p, h->vals.p, h->vals2.p, n, bg.passes, h->gd.p, s, &in_b));
  const uint32_t* keys_sorted = in_b ? h->keys2.p : h->keys.p;
  const uint32_t* vals_sorted = in_b ? h->vals2.p : h->vals.p;
  HIP_TRY(h, launch_find_runs(keys_sorted, n, h->gd.p, h->gdh.d, /*min_pts=*/1, h->nleaf.p, h->run_counts.p, h->run_offsets.p,
                              h->tickets.p + 1, nullptr, 0, &h->run_seq, h->leaf_start.p, h->leaf_cnt.p, s));
  launch_voxel_centroids(h->xyz4.p, di, vals_sorted, h->nleaf.p, h->leaf_start.p, h->leaf_cnt.p, n, cap, ox, oy, oz, oi, s);
  HIP_TRY(h, hipGetLastError());
  HIP_TRY(h, hipMemcpyAsync(h->small.h + 12, h->nleaf.p, sizeof(int), hipMemcpyDeviceToHost, s));
  HIP_TRY(h, hipStreamSynchronize(s));
  *n_out = (size_t)h->small.h[12];
  if (*n_out > cap) return fail(h, NDT_ERR_INVALID_ARG, "output capacity too small: " + std::to_string(*n_out) + " occupied voxels");
  return NDT_OK;
}

int ndt_voxel_downsample_device(ndt_handle* h, const float* dx, const float* dy, const float* dz, const float* d_intensity,
                                size_t n, float leaf, float* ox, float* oy, float* oz, float* o_intensity, size_t cap,
                                size_t* n_out) {
  if (!h || !n_out || ((!dx || !dy || !dz) && n) || ((!ox || !oy || !oz) && cap) || !(leaf > 1e-6f)) return NDT_ERR_INVALID_ARG;
  int rc = bind_device(h);
  if (rc) return rc;
  return voxel_downsample_device_impl(h, dx, dy, dz, d_intensity, n, leaf, ox, oy, oz, o_intensity, cap, n_out);
}

int ndt_voxel_downsample(ndt_handle* h, const float* xyz, size_t n, size_t stride_bytes, long intensity_offset_bytes,
                         float leaf, float* out, size_t cap, size_t* n_out) {
  if (!h || !n_out || (!xyz && n) || (!out && cap) || stride_bytes < 12 || stride_bytes % 4 || !(leaf > 1e-6f) ||
      (intensity_offset_bytes >= 0 && (intensity_offset_bytes % 4 || (size_t)intensity_offset_bytes + 4 > stride_bytes ||
                                       intensity_offset_bytes < 12)))
    return NDT_ERR_INVALID_ARG;
  int rc = bind_device(h);
  if (rc) return rc;
  *n_out = 0;
  if (n == 0) return NDT_OK;
  const bool has_i = intensity_offset_bytes >= 0;
  DevBuf<float> x, y, z, in_i, o;   // scratch of this call (a shutdown-time operation in the reference)
  auto done = [&](int code) { x.release(); y.release(); z.release(); in_i.release(); o.release(); return code; };
  rc = upload_soa(h, h->lane_t, h->stream, xyz, nullptr, nullptr, nullptr, n, stride_bytes, x, y, z, true);
  if (rc) return done(rc);
  std::vector<float> tmp;
  if (has_i) {
    tmp.resize(n);
    const char* base = reinterpret_cast<const char*>(xyz) + intensity_offset_bytes;
    for (size_t i = 0; i < n; ++i) tmp[i] = *reinterpret_cast<const float*>(base + i * stride_bytes);
    hipError_t e = in_i.ensure(n);
    if (e == hipSuccess) e = hipMemcpy(in_i.p, tmp.data(), n * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) return done(fail(h, NDT_ERR_HIP, hipGetErrorString(e)));
  }
  const size_t ocap = std::min(cap, n);
  hipError_t e = o.ensure(4 * std::max<size_t>(ocap, 1));
  if (e != hipSuccess) return done(fail(h, e == hipErrorOutOfMemory ? NDT_ERR_ALLOC : NDT_ERR_HIP, hipGetErrorString(e)));
  float* ox = o.p, *oy = o.p + ocap, *oz = o.p + 2 * ocap, *oi = o.p + 3 * ocap;
  rc = voxel_downsample_device_impl(h, x.p, y.p, z.p, has_i ? in_i.p : nullptr, n, leaf, ox, oy, oz, has_i ? oi : nullptr, ocap, n_out);
  if (rc) return done(rc);
  const size_t m = *n_out;
  std::vector<float> back(4 * ocap);
  e = hipMemcpy(back.data(), o.p, (has_i ? 4 : 3) * ocap * sizeof(float), hipMemcpyDeviceToHost);
  if (e != hipSuccess) return done(fail(h, NDT_ERR_HIP, hipGetErrorString(e)));
  char* ob = reinterpret_cast<char*>(out);
  for (size_t i = 0; i < m; ++i) {
    float* p = reinterpret_cast<float*>(ob + i * stride_bytes);
    p[0] = back[i]; p[1] = back[ocap + i]; p[2] = back[2 * ocap + i];
    if (has_i) *reinterpret_cast<float*>(ob + i * stride_bytes + intensity_offset_bytes) = back[3 * ocap + i];
  }
  return done(NDT_OK);
}

int ndt_set_global_source_size(ndt_handle* h, int64_t n_total) {
  if (!h) return NDT_ERR_INVALID_ARG;
  h->n_src_global = n_total;
  return NDT_OK;
}

int ndt_set_regularization_pose(ndt_handle* h, const float pose[16]) {
  if (!h || !pose) return NDT_ERR_INVALID_ARG;
  std::memcpy(h->reg_pose, pose, sizeof(h->reg_pose));
  h->have_reg = true;
  return NDT_OK;
}

int ndt_clear_regularization_pose(ndt_handle* h) {
  if (!h) return NDT_ERR_INVALID_ARG;
  h->have_reg = false;
  return NDT_OK;
}

int ndt_set_keepwarm(ndt_handle* h, int period_us) {
  if (!h || period_us < 0 || (period_us > 0 && period_us < 100)) return NDT_ERR_INVALID_ARG;
  h->keepwarm.start(h->device, h->n_cus, period_us);
  return NDT_OK;
}
int ndt_get_keepwarm(const ndt_handle* h, long long* beats) {
  if (!h) return NDT_ERR_INVALID_ARG;
  if (beats) *beats = h->keepwarm.beats();
  return h->keepwarm.period_us();
}

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
  EvalFn fn = [h](const double* p, const float* T, bool need_h, Eval* e) { return evaluate(h, p, T, need_h, e); };
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
      const double us = 1e3 * out->ms_total / (double)launched;
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

int ndt_get_grid_info(const ndt_handle* hc, ndt_grid_info* out) {
  if (!hc || !out) return NDT_ERR_INVALID_ARG;
  std::memset(out, 0, sizeof(*out));
  ndt_handle* h = const_cast<ndt_handle*>(hc);  // logically const: a pending build is the grid it describes
  if (h->build_pending || h->src_upload_pending) {
    int rc = bind_device(h);
    if (rc) return rc;
  }
  {
    int rc = settle(h);
    if (rc) return rc;
  }
  if (!h->have_grid) return NDT_ERR_NO_TARGET;
  for (int a = 0; a < 3; ++a) {
    out->min_b[a] = h->geom.min_b[a];
    out->max_b[a] = h->max_b[a];
    out->div_b[a] = h->geom.div_b[a];
  }
  out->leaf_size = h->geom.leaf;
  out->inverse_leaf_size = h->geom.inv_leaf;
  out->n_leaves = h->n_valid;
  out->n_cells = h->geom.ncells;
  out->n_target_points = (int64_t)h->n_tgt;
  out->ms_build = h->ms_build;
  return NDT_OK;
}

int64_t ndt_export_leaves(ndt_handle* h, ndt_leaf* out, size_t cap) {
  if (!h || (!out && cap)) return NDT_ERR_INVALID_ARG;
  if (bind_device(h)) return NDT_ERR_HIP;
  {
    int rc = settle(h);
    if (rc) return rc;
  }
  if (!h->have_grid) return NDT_ERR_NO_TARGET;
  std::vector<LeafStats> st((size_t)h->n_slots);
  if (h->multi_active) {
    st = h->multi_stats;  // table order; `cell` is the union grid's index
  } else if (h->n_slots) {
    hipError_t e = hipMemcpy(st.data(), h->stats.p, st.size() * sizeof(LeafStats), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return fail(h, NDT_ERR_HIP, hipGetErrorString(e));
  }
  std::vector<const LeafStats*> ok;
  ok.reserve(st.size());
  for (const auto& L : st)
    if (L.count > 0) ok.push_back(&L);
  std::stable_sort(ok.begin(), ok.end(), [](const LeafStats* a, const LeafStats* b) { return a->cell < b->cell; });
  const size_t n = std::min(cap, ok.size());
  const GridGeom& g = h->geom;
  for (size_t i = 0; i < n; ++i) {
    const LeafStats& L = *ok[i];
    ndt_leaf& o = out[i];
    o.index = L.cell;
    o.point_count = L.count;
    const int i0 = L.cell % g.div_b[0], i1 = (L.cell / g.div_b[0]) % g.div_b[1], i2 = L.cell / g.mul2;
    o.center[0] = ((float)(g.min_b[0] + i0) + 0.5f) * g.leaf;
    o.center[1] = ((float)(g.min_b[1] + i1) + 0.5f) * g.leaf;
    o.center[2] = ((float)(g.min_b[2] + i2) + 0.5f) * g.leaf;
    std::memcpy(o.mean, L.mean, sizeof(o.mean));
    std::memcpy(o.cov, L.cov, sizeof(o.cov));
    std::memcpy(o.icov, L.icov, sizeof(o.icov));
    std::memcpy(o.evecs, L.evecs, sizeof(o.evecs));
    std::memcpy(o.evals, L.evals, sizeof(o.evals));
  }
  return (int64_t)n;
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

void ndt_shard_range(size_t n, int rank, int nranks, size_t* begin, size_t* count) {
  if (nranks < 1) nranks = 1;
  if (rank < 0) rank = 0;
  if (rank >= nranks) rank = nranks - 1;
  const size_t b = n * (size_t)rank / (size_t)nranks, e = n * (size_t)(rank + 1) / (size_t)nranks;
  if (begin) *begin = b;
  if (count) *count = e - b;
}

int ndt_comm_unique_id(void* out128) {
  if (!out128) return NDT_ERR_INVALID_ARG;
  return Reducer::unique_id(out128);
}

int ndt_comm_init_rccl(ndt_handle* h, const void* id128, int rank, int nranks) {
  if (!h || !id128) return NDT_ERR_INVALID_ARG;
  int rc = bind_device(h);
  if (rc) return rc;
  return h->red.init_rccl(id128, rank, nranks, &h->err);
}

int ndt_comm_init_shm(ndt_handle* h, const char* name, int rank, int nranks) {
  if (!h) return NDT_ERR_INVALID_ARG;
  return h->red.init_shm(name, rank, nranks, &h->err);
}

int ndt_comm_p2p_handle(ndt_handle* h, void* out_handle) {
  if (!h || !out_handle) return NDT_ERR_INVALID_ARG;
  int rc = bind_device(h);
  if (rc) return rc;
  return h->red.p2p_handle(out_handle, &h->err);
}

int ndt_comm_init_p2p(ndt_handle* h, const void* handles, int rank, int nranks) {
  if (!h || !handles) return NDT_ERR_INVALID_ARG;
  int rc = bind_device(h);
  if (rc) return rc;
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return h->red.init_p2p(handles, rank, nranks, &h->err);
}

int ndt_comm_p2p_stats(ndt_handle* h, int64_t out[4], int reset) {
  if (!h || !out) return NDT_ERR_INVALID_ARG;
  int rc = bind_device(h);
  if (rc) return rc;
  unsigned long long v[4] = {0, 0, 0, 0};
  rc = h->red.p2p_stats(v, reset != 0, &h->err);
  for (int k = 0; k < 4; ++k) out[k] = (int64_t)v[k];
  return rc;
}

int ndt_comm_p2p_selftest(ndt_handle* h, int rounds, int64_t out[4]) {
  if (!h || !out || rounds < 1) return NDT_ERR_INVALID_ARG;
  int rc = bind_device(h);
  if (rc) return rc;
  quit_prelaunched(h);
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream2));
  unsigned long long v[4] = {0, 0, 0, 0};
  rc = h->red.p2p_selftest(rounds, v, &h->err);
  for (int k = 0; k < 4; ++k) out[k] = (int64_t)v[k];
  return rc;
}

int ndt_comm_init_hook(ndt_handle* h, ndt_allreduce_fn fn, void* ctx, int rank, int nranks) {
  if (!h) return NDT_ERR_INVALID_ARG;
  return h->red.init_hook(fn, ctx, rank, nranks);
}

int ndt_comm_destroy(ndt_handle* h) {
  if (!h) return NDT_ERR_INVALID_ARG;
  h->red.destroy();
  return NDT_OK;
}

int ndt_result_covariance(const double hessian36[36], double eps, int gtsam_order, double cov36[36]) {
  if (!hessian36 || !cov36 || !std::isfinite(eps)) return NDT_ERR_INVALID_ARG;
  return result_covariance(hessian36, eps, gtsam_order != 0, cov36) ? NDT_OK : NDT_ERR_INVALID_ARG;
}

// test seam (not in the public header): evaluations served by a pre-launched kernel, pre-launched
// kernels told to leave, mailbox time-outs
int ndt_debug_prelaunch_counters(const ndt_handle* h, int64_t out[8]) {
  if (!h || !out) return NDT_ERR_INVALID_ARG;
  out[0] = h->n_prelaunch_used; out[1] = h->n_prelaunch_quit; out[2] = h->n_prelaunch_timeouts;
  out[3] = h->n_prelaunch_overlapped;
  out[4] = h->n_p2p_host_finishes;
  out[5] = h->n_lost_row_retries;
  out[6] = h->auto_one_stream ? 1 : 0;   // the stream placement NDT_PRELAUNCH_AUTO has settled on (1: one stream)
  out[7] = h->n_auto_switches;
  return NDT_OK;
}

// tuning aid (not in the public header): the first-evaluation short cut on / off per handle
int ndt_debug_set_speculation(ndt_handle* h, int on) {
  if (!h) return NDT_ERR_INVALID_ARG;
  h->spec_enabled = on != 0;
  return NDT_OK;
}

// test seam (not in the public header): first evaluations enqueued behind a deferred build {kept, discarded}
int ndt_debug_speculation_counters(const ndt_handle* h, int64_t out[2]) {
  if (!h || !out) return NDT_ERR_INVALID_ARG;
  out[0] = h->n_spec_used;
  out[1] = h->n_spec_discarded;
  return NDT_OK;
}

// test seam (not in the public header): builds that fell back from the fused sort passes to the
// classic ones after a block gave up waiting
int ndt_debug_build_counters(const ndt_handle* h, int64_t out[3]) {
  if (!h || !out) return NDT_ERR_INVALID_ARG;
  out[0] = h->n_fused_sort_fallbacks;
  out[1] = h->n_bucket_fallbacks;   // two-launch builds that were declined and repeated sort-based
  out[2] = h->n_bucket_builds;      // builds that went through in two launches
  return NDT_OK;
}

// diagnostic builds only (-DNDT_STAMPS): 8 x 100 MHz stamps per block of the last launch
int ndt_debug_read_stamps(unsigned long long* out, int nblocks) { return derivs_read_stamps(out, nblocks); }
int ndt_debug_read_build_stamps(unsigned long long* out) { return build_read_stamps(out); }
int ndt_debug_read_wave_stamps(unsigned long long* out, int nblocks) { return derivs_read_wave_stamps(out, nblocks); }
// test seam (not in the public header): the finishing-wave tables of a block shape (k_derivatives)
int ndt_debug_item_owners(int threads, unsigned int* owners, unsigned int* fin_waves) {
  if (threads < 64 || threads > 1024 || threads % 64 != 0 || !owners || !fin_waves) return NDT_ERR_INVALID_ARG;
  derivs_item_owners(threads, owners, fin_waves);
  return NDT_OK;
}

// test seam (not in the public header): the voxel build's radix sort on caller-supplied keys;
// vals_out receives the stable sorting permutation.  Host arrays.
int ndt_debug_sort_pairs(ndt_handle* h, const uint32_t* keys, size_t n, int end_bit, uint32_t* keys_out,
                         uint32_t* vals_out) {
  if (!h || (n && (!keys || !keys_out || !vals_out)) || end_bit < 1 || end_bit > 32) return NDT_ERR_INVALID_ARG;
  if (n > (size_t)std::numeric_limits<int>::max() / 2) return NDT_ERR_INVALID_ARG;
  if (n == 0) return NDT_OK;
  int rc = bind_device(h);
  if (rc) return rc;
  settle_discard(h);
  hipStream_t s = h->stream;
  HIP_TRY(h, h->keys.ensure(n));
  HIP_TRY(h, h->vals.ensure(n));
  HIP_TRY(h, h->keys2.ensure(n));
  HIP_TRY(h, h->vals2.ensure(n));
  HIP_TRY(h, h->sort_tmp.ensure(sort_temp_bytes(n)));
  HIP_TRY(h, h->gd.ensure(1));
  BuildGeom plan{};
  fill_sort_plan(&plan, end_bit);
  plan.status = BG_OK;
  HIP_TRY(h, hipMemcpyAsync(h->gd.p, &plan, sizeof(plan), hipMemcpyHostToDevice, s));
  HIP_TRY(h, hipMemcpyAsync(h->keys.p, keys, n * sizeof(uint32_t), hipMemcpyHostToDevice, s));
  launch_sort_first_count(h->keys.p, n, h->gd.p, h->sort_tmp.p, s);
  bool in_b = false;
  HIP_TRY(h, sort_pairs(h->sort_tmp.p, h->keys.p, h->keys2.p, h->vals.p, h->vals2.p, n, plan.passes, h->gd.p, s, &in_b));
  HIP_TRY(h, hipMemcpyAsync(keys_out, in_b ? h->keys2.p : h->keys.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  HIP_TRY(h, hipMemcpyAsync(vals_out, in_b ? h->vals2.p : h->vals.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  HIP_TRY(h, hipStreamSynchronize(s));
  h->have_grid = false;  // the build's scratch was overwritten
  return NDT_OK;
}

int ndt_set_handoff_mode(ndt_handle* h, int mode) {
  if (!h) return NDT_ERR_INVALID_ARG;
  if (mode != NDT_HANDOFF_ASYNC && mode != NDT_HANDOFF_SYNC) return fail(h, NDT_ERR_INVALID_ARG, "unknown hand-off mode");
  h->handoff_mode = mode;
  return NDT_OK;
}

int ndt_get_handoff_mode(const ndt_handle* h) { return h ? h->handoff_mode : NDT_ERR_INVALID_ARG; }

int ndt_wait(ndt_handle* h) {
  if (!h) return NDT_ERR_INVALID_ARG;
  int rc = bind_device(h);
  if (rc) return rc;
  (void)settle_build(h);
  rc = settle_source(h);
  if (rc) return rc;
  rc = lane_wait(h, h->lane_t);
  if (rc) return rc;
  rc = lane_wait(h, h->lane_s);
  if (rc) return rc;
  if (!h->have_grid && h->deferred_rc) return report_deferred(h);
  return NDT_OK;
}

int ndt_get_handoff_timing(const ndt_handle* h, ndt_handoff_timing* out) {
  if (!h || !out) return NDT_ERR_INVALID_ARG;
  std::memset(out, 0, sizeof(*out));
  out->target = h->lane_t.tm;
  out->source = h->lane_s.tm;
  out->ms_build_wait = h->ms_settle_wait;
  out->mode = h->handoff_mode;
  out->cpu_budget = host_cpu_budget();
  out->repack_workers = (int)repack_workers();
  return NDT_OK;
}

int ndt_enable_kernel_timing(ndt_handle* h, int on) {
  if (!h) return NDT_ERR_INVALID_ARG;
  h->timing = on != 0;
  return NDT_OK;
}

int ndt_get_timing(const ndt_handle* h, ndt_timing* out) {
  if (!h || !out) return NDT_ERR_INVALID_ARG;
  *out = h->tm;
  return NDT_OK;
}

}  // extern "C"
