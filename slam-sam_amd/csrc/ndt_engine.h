// ndt_engine.h -- the engine's handle and the functions its translation units share (internal; the public
// boundary is include/ndt_hip.h).  Round 5 split the former ndt_api.hip along these lines:
//   ndt_handle.hip     handle life cycle, parameters, reducers' entry points, timing, test seams
//   ndt_handoff.hip    host hand-off (repack + pull kernels), the target voxel-grid build's orchestration, grid accessors
//   ndt_evaluate.hip   derivative evaluations (ordinary, pre-launched, batched), align, scoring
//   ndt_keyframes.hip  multi-grid targets, the device-resident keyframe archive, voxel downsample
// One handle = one engine instance = one HIP stream on one gfx950 device; it owns every device allocation.  There is no
// CPU path: without a device every compute call fails with NDT_ERR_NO_DEVICE.
#pragma once

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

namespace ndt {
namespace engine {

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

}  // namespace engine
}  // namespace ndt

using namespace ndt;
using namespace ndt::engine;

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
  DevBuf<uint32_t> bucket_tab;        // two-launch build: where a tile holds a bucket's points, [bucket][tile] = {count : 16 | first : 16}
  DevBuf<int> bnd;                    // its 8 bounds words {min xyz, max xyz, #finite, largest bucket}; neutral between builds
  long long n_bucket_builds = 0, n_bucket_fallbacks = 0;
  long long n_chunked_pass_builds = 0, n_chunked_pass_launches = 0;   // host hand-offs whose partition ran under the transfer
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
  hipStream_t pstream = nullptr;     // partition launches of a host target's build, beside the pull kernels on `stream` (created on first use)
  std::vector<hipEvent_t> pass_ev;   // ... one event per chunk in flight + the join
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
    // the asynchronous host hand-off runs the two-launch build's partition under the transfer, chunk by chunk
    bool pass_chunked = false;   // attempt 0 finds its partition enqueued already (every tile)
    int pass_tiles = 0;          // tiles enqueued so far
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


#define HIP_TRY(h, expr)                                                                  \
  do {                                                                                    \
    hipError_t e__ = (expr);                                                              \
    if (e__ != hipSuccess)                                                                \
      return fail(h, (e__ == hipErrorOutOfMemory) ? NDT_ERR_ALLOC : NDT_ERR_HIP,          \
                  std::string(#expr) + ": " + hipGetErrorString(e__));                    \
  } while (0)

namespace ndt {
namespace engine {

int fail(ndt_handle* h, int code, const std::string& msg);
int bind_device(ndt_handle* h);
bool params_valid(const ndt_params* p, std::string* why);
int lane_wait(ndt_handle* h, ndt_handle::UploadLane& lane);
int upload_soa(ndt_handle* h, ndt_handle::UploadLane& lane, hipStream_t stream, const float* xyz, const float* x,
               const float* y, const float* z, size_t n, size_t stride, DevBuf<float>& dx, DevBuf<float>& dy,
               DevBuf<float>& dz, bool sync, const std::function<void(size_t /* points on the device so far */, bool /* last chunk */)>* after_chunk = nullptr);
bool timing_brackets_launch();
bool auto_probe_enabled();
int pack_records(ndt_handle* h, bool wait);
int records_for_eval(ndt_handle* h, EvalConsts* ec, const VoxelRecord** rec);
int neutral_bounds(ndt_handle* h);
int build_begin(ndt_handle* h, const float* x, const float* y, const float* z, size_t n, ndt_handle::BuildRun& br);
int build_enqueue(ndt_handle* h, ndt_handle::BuildRun& br);
int build_collect(ndt_handle* h, ndt_handle::BuildRun& br);
int build_complete(ndt_handle* h, ndt_handle::BuildRun& br);
int build_grid(ndt_handle* h, const float* x, const float* y, const float* z, size_t n, bool defer = false);
int settle_build(ndt_handle* h);
int settle_source(ndt_handle* h);
int source_behind_target_transfer(ndt_handle* h, bool async);
int report_deferred(ndt_handle* h);
int settle(ndt_handle* h);
void settle_discard_keep_grid(ndt_handle* h);
void settle_discard(ndt_handle* h);
void fill_pose_consts(const double p[6], const float T[16], PoseConsts* pc);
unsigned long long process_item_salt();
EvalConsts make_eval_consts(const ndt_handle* h, bool need_h);
int ensure_counters(ndt_handle* h, size_t k);
int ready_for_eval(ndt_handle* h);
int ensure_flag_slots(ndt_handle* h, size_t K);
int maybe_sort_source(ndt_handle* h, const float T[16]);
bool source_sort_wanted(const ndt_handle* h, int n_valid);
bool first_eval_behind_build(const ndt_handle* h);
int ensure_partials(ndt_handle* h, size_t words);
bool slots_complete(const volatile unsigned long long* slots, unsigned long long seq);
int wait_slots(ndt_handle* h, unsigned long long seq, int K = 1, int first = 0);
bool ensure_mailbox(ndt_handle* h);
void publish_pose(ndt_handle* h, unsigned long long seq, const PoseConsts& pc);
void quit_prelaunched(ndt_handle* h);
int evaluate(ndt_handle* h, const double p[6], const float T[16], bool need_h, Eval* out, bool score_only = false,
             bool safe_retry = false);

}  // namespace engine
}  // namespace ndt
