// ndt_device.h -- structures shared by the host code and the HIP kernels of the
// NDT engine (gfx950 only).  Not part of the public ABI (that is include/ndt_hip.h).
#pragma once
#include <cstddef>

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ndt {

// Dense-index voxel grid geometry.  Everything a lookup needs, computed on the
// host in f32 exactly as the reference does (ref:
// extern/svn_ndt/include/voxel_grid_covariance_impl.hpp:129-140 for min_b /
// div_b / divb_mul, :59-64 for the f32 bounds).
struct GridGeom {
  float leaf;
  float inv_leaf;
  int min_b[3];
  int div_b[3];
  float lo[3];  // (float)min_b * leaf
  float hi[3];  // (float)(max_b + 1) * leaf
  int mul1;     // div_b[0]
  int mul2;     // div_b[0] * div_b[1]
  int ncells;   // div_b[0] * div_b[1] * div_b[2]
};

// Geometry + sort plan of one target build, computed ON THE DEVICE by the last block of the
// bounds kernel (the host used to stop mid-build for it) and read by every later build kernel
// from device memory; the host gets a copy through pinned memory at the end of the build.
enum { BG_OK = 0, BG_NO_FINITE = 1, BG_OVERFLOW = 2, BG_CAPACITY = 3, BG_PASSES = 4, BG_SPIN = 5,
       BG_BUCKET = 6 /* the two-launch bucketed build declined this cloud (a bucket beyond a block's LDS, huge
                        coordinates): the host repeats the build with the sort-based pipeline */ };
struct BuildGeom {
  GridGeom g;
  int max_b[3];
  int bits;       // bits of a cell key, the sentinel `ncells` included
  int passes;     // digit passes of the radix sort
  int width[4];   // bits per pass
  int shift[4];   // first bit of each pass
  int n_finite;
  int status;     // BG_*
};

// What the derivative kernel reads per (point, voxel) pair: 80 B, five 16-B
// loads.  mean + upper triangle of the inverse covariance in f64 -- the
// reference keeps both in f64 and forms the Mahalanobis distance in f64
// (ref: svn_ndt_impl.hpp:418, voxel_grid_covariance.h:125-128).
// (Padding a record to one 128-byte line was measured in round 3, when the two-launch build stopped handing out leaf
// slots in cell order: no gain, profiles/r03_step_ab_record_order.txt.)
struct alignas(16) VoxelRecord {
  double mean[3];
  double icov[6];  // xx, xy, xz, yy, yz, zz
  double pad;
};

// Full per-leaf statistics kept for export (ref: voxel_grid_covariance.h:99-131).
struct LeafStats {
  int32_t cell;
  int32_t count;  // < 0: rejected by the eigenvalue / inverse checks
  double mean[3];
  double cov[9];
  double icov[9];
  double evecs[9];
  double evals[3];
};

// The same record packed into 48 bytes: the mean stays f64 (x' - mu loses nothing), the inverse
// covariance is rounded to f32 -- what the reference's own per-pair gradient / Hessian arithmetic does with it
// (c_inv4, svn_ndt_impl.hpp:449-456); only the score's Mahalanobis term sees an f32 matrix where the reference keeps
// f64 (6e-8 relative).  Three 16-byte loads per neighbour instead of five.  ndt_set_record_format().
struct alignas(16) PackedRecord {
  double mean[3];
  float icov[6];  // xx xy xz yy yz zz
};
static_assert(sizeof(PackedRecord) == 48, "three 16-byte loads");

// Pose-dependent constants of one derivative evaluation.  Passed by value as a
// kernel argument (K = 1) or read from a device array (pose batches).
struct PoseConsts {
  float R[9];      // row-major rotation, f32
  float t[3];
  float jang[24];  // 8x3  (ref: svn_ndt_impl.hpp:271-290)
  float hang[45];  // 15x3 (ref: svn_ndt_impl.hpp:299-331)
};

// Mailbox of a PRE-LAUNCHED evaluation kernel (round 2): the kernel of evaluation N + 1 is put on
// the stream while evaluation N is still running, starts the moment N has finished and waits,
// one lane per block, for the host to publish the pose here -- fine-grained device memory the
// host writes through the PCIe BAR.  words = the 81 floats of a PoseConsts; seq is written last.
// seq == the kernel's own sequence number: go; seq == that number | MBOX_QUIT: leave at once.
constexpr int MBOX_GRANULES = 82;  // 81 pose words + one control word
constexpr unsigned int MBOX_CTRL_QUIT = 0x51554954u;  // "QUIT"
struct PoseMailbox {
  // plain form (NDT_MBOX_TAGGED=0): pose words first, sequence number last
  unsigned long long seq;
  unsigned int pad[2];
  unsigned int words[84];
  unsigned int pad2[40];
  // tagged form: 82 self-validating 8-byte granules {32-bit launch tag, 32-bit word} -- granule k < 81
  // carries pose word k, granule 81 the control word (0, or MBOX_CTRL_QUIT).  The host writes every
  // granule with ONE aligned 64-bit store: the largest store that the compiler, x86 and a partially
  // flushed write-combining buffer are all certain to keep whole (a first version used 16-byte slots
  // written through _mm_store_si128).  Lane k < 41 of wave 0 reads granules 2k and 2k + 1 with one
  // 16-byte load and checks both tags.
  unsigned int gran[MBOX_GRANULES][2];
};
static_assert(offsetof(PoseMailbox, gran) == 512, "granules are 16-byte aligned in pairs");
// the 32-bit tag of a launch: odd (never the 0 of a fresh mailbox), unique for 2^31 launches
__host__ __device__ inline unsigned int mbox_tag32(unsigned long long seq) { return ((unsigned int)seq << 1) | 1u; }
constexpr unsigned long long MBOX_QUIT = 1ull << 63;
constexpr unsigned long long MBOX_TIMEOUT_TICKS = 2000000ull;  // 20 ms of the 100 MHz s_memrealtime clock

constexpr int SUMMER_SPLIT = 4;   // summing blocks of a single-pose launch where compute units are spare (k_derivatives)

struct EvalConsts {
  double d1, d2;   // Gauss constants (ref: svn_ndt_impl.hpp:80-131)
  int direct7;     // 1: centre + 6 face neighbours, 0: centre only (ignored in KDTREE mode)
  int kdtree;      // 1: radius search over voxel centroids (27-cell scan)
  float kd_radius2;  // (float)(radius^2), radius = leaf size (ref: svn_ndt_impl.hpp:579)
  int need_hessian;
  int gauss_newton;
  int single_level_max;  // grids with more rows than this take the two-level final sum
  int direct26;    // 1: every valid voxel of the 3x3x3 block around the point's cell (pclomp DIRECT26)
  int score_only;  // 1: score / NVTL / counts only, no gradient or Hessian (ndt_score_transform)
  int fixed_summer;  // 1: block 0 adds the partial rows (polls their tags), no tickets (single-level grids)
  int dedicated_summer;  // n > 0: blocks 0 .. n - 1 of the grid own no points: they only add the rows -- 1 (all 32 words), or
                         // SUMMER_SPLIT (eight words = one 128-byte line of every row each; single-pose launches)
  int doubling_split;    // 1: no dedicated summing block, and the blocks of rows 0 .. SUMMER_SPLIT - 1 add eight words each behind their own points
  int mbox_tagged;  // pre-launched kernels: 1 = the pose arrives as tagged 8-byte granules, 0 = words then sequence number
  int mbox_preload; // pre-launched kernels: 1 = the point is fetched before the wait for the pose
  int multigrid;   // 1: the table is a union of grids (radius search, leaves of a cell chained through VoxelRecord::pad)
  int packed;      // 1: the kernel's `rec` argument points at PackedRecord[] (48 bytes per leaf) instead of VoxelRecord[]
  int xcd_count;   // > 1: the blocks an XCD receives (linear workgroup id mod xcd_count) take CONSECUTIVE chunks of the source
  int xcd_stripe;  // ... per stripe of xcd_count * xcd_stripe blocks (0: the whole row is one stripe); see xcd_chunk()
  int compute_units;  // host side only: compute units of the handle's device (block shapes, XCD count)
  int own_units;      // host side only: 1 = the handle has its device to itself (no multi-rank reducer): one block per compute unit
  int safe_sum;    // 1: the final sum is made by the block that draws the LAST TICKET (no block waits for rows of blocks that
                   // may not be resident): the re-evaluation after a lost row
  unsigned int item_owner;  // k_derivatives: 2 bits per wave of a block -- the SIMD whose finishing wave expands that wave's points
  unsigned long long item_salt;  // ... per-process random bits in the tags the waves of a block publish their items under (process_item_salt())
  int lone_wave;            // ... a finishing wave that expands nobody but itself (-1: none)
  unsigned int fin_waves;   // ... 4 bits per SIMD: its finishing wave (derivs_item_owners(), ndt_derivs.hip; per launch, from the block shape)
  int mute_row;    // test seam (libndt_hip_seams.so only): row + 1 of the block that withholds its partial row; 0 = none
};

// Which chunk of the source a block works on.  The hardware hands workgroup w (linear id over the whole grid) to XCD
// w mod X, and every XCD has its own 4 MB L2 that is cold at the start of a launch: with chunk = block id each L2 sees
// every X-th stretch of the scan, i.e. the whole map, and fetches the whole record table and index grid once per XCD.
// With the blocks of one XCD on CONSECUTIVE chunks an L2 serves one X-th of a stretch of the scan.
//   p    position of the block among the L blocks that share the work (point blocks of one pose's row)
//   off  XCD of position 0 (a row of a batched launch starts wherever the previous row ended; a dedicated summing
//        block in front shifts everything by one)
//   m    chunks per XCD and STRIPE, 0 = the whole row is one stripe.  A grid of several residency rounds is cut into
//        stripes of X * m positions (one round each): within a stripe XCD x takes m consecutive chunks, so every XCD
//        works through every stripe (an XCD that drew an expensive eighth of the WHOLE source finished late, round
//        after round: round 3's reason to keep this to resident grids) and still sees one X-th of the stripe.
// A bijection of [0, L) for any L, off, X >= 1, m; rows are numbered by chunk, so the final sum -- fixed order over the
// rows -- does not change by a bit whatever the mapping.
__host__ __device__ inline int xcd_chunk(int p, int L, int off, int X, int m) {
  int base = 0, l = p, len = L;
  if (m > 0) {
    const int S = X * m;
    base = (p / S) * S;
    l = p - base;
    len = (L - base < S) ? L - base : S;
  }
  const int x = (l + off) % X;              // the XCD this block runs on
  const int b0 = (x - off % X + X) % X;     // first position of the stripe on that XCD
  const int j = (l - b0) / X;               // rank among the stripe's blocks of that XCD
  const int q = len / X, r = len % X;
  int start = 0;
  for (int xx = 0; xx < x; ++xx) start += q + (((xx - off % X + X) % X) < r ? 1 : 0);
  return base + start + j;
}

// layout of one evaluation (matches NDT_EVAL_WORDS in include/ndt_hip.h)
enum {
  EV_SCORE = 0,
  EV_G = 1,        // 6
  EV_H = 7,        // 21, upper triangle row by row
  EV_NVTL = 28,
  EV_NWITH = 29,
  EV_NPAIRS = 30,
  EV_FAIL = 31,    // 0; 1 = the in-kernel final sum gave up waiting for a partial row,
                   // 2 = a pre-launched kernel gave up waiting for its pose (nothing was evaluated),
                   // 3 = NDT_REDUCE_P2P: this rank's sum was published, a peer's row had not arrived after
                   //     XCHG_TIMEOUT_TICKS (the host finishes the exchange itself)
  EV_WORDS = 32
};

// ---- cross-GPU sum of an evaluation INSIDE the kernel's final sum (NDT_REDUCE_P2P) -------------------
// Every rank owns an exchange area in fine-grained device memory that all ranks of the node have mapped
// (hipIpcOpenMemHandle; xGMI peer access).  The block that finishes a rank's local sum writes its 32
// tagged 16-byte slots {round, value} into EVERY rank's area (row = its own rank), then polls the N rows of
// its own area until all carry this round's tag, adds them in rank order and hands the GLOBAL evaluation
// to its host -- one one-shot all-gather + local sum, no second launch, no host
// hop, and the pre-launched / host-polled fast path of a single GPU stays on.  The round number is the
// count of global evaluations: every rank runs the identical host loop on identical sums, so the counts
// agree without being exchanged.  Two generations of rows (round parity): a rank can be at most one round
// ahead of the slowest -- it cannot publish round k + 2 before every peer has published k + 1, which a
// peer does only after it has finished reading round k (the lesson of the alternating result buffers).
constexpr int XCHG_MAX_RANKS = 64;
constexpr size_t XCHG_SINGLE_BYTES = (size_t)2 * XCHG_MAX_RANKS * EV_WORDS * 16;  // 64 KB: rows of single-pose evaluations
// Behind them, in the same allocation (one IPC handle): rows of BATCHED evaluations (ndt_eval_derivatives with K poses,
// SVN Stage 1), exchanged by the host in ONE round per batch of up to XCHG_BATCH_MAX poses: [generation][pose][rank][32 slots]
constexpr int XCHG_BATCH_MAX = 64;
constexpr size_t XCHG_BATCH_BYTES = (size_t)2 * XCHG_BATCH_MAX * XCHG_MAX_RANKS * EV_WORDS * 16;  // 4 MB
constexpr size_t XCHG_AREA_BYTES = XCHG_SINGLE_BYTES + XCHG_BATCH_BYTES;
struct XchgInfo {
  int rank, nranks;
  unsigned long long area[XCHG_MAX_RANKS];  // every rank's exchange area as mapped in THIS process; [rank] is local
  // four counters in this rank's device memory (ndt_comm_p2p_stats): exchanges made inside a kernel, their summed and
  // their longest duration -- own row published to every row gathered -- in 10 ns ticks, exchanges a peer was late for
  unsigned long long stats;
};
__host__ __device__ inline unsigned int xchg_slot_offset(unsigned long long round, int row, int word) {
  return (unsigned int)((((unsigned int)(round & 1ull) * XCHG_MAX_RANKS + (unsigned int)row) * EV_WORDS + (unsigned int)word) * 16u);
}
__host__ __device__ inline size_t xchg_batch_offset(unsigned long long round, int pose, int row, int word) {
  return XCHG_SINGLE_BYTES + ((((size_t)(round & 1ull) * XCHG_BATCH_MAX + (size_t)pose) * XCHG_MAX_RANKS + (size_t)row) * EV_WORDS + (size_t)word) * 16u;
}
constexpr unsigned long long XCHG_TIMEOUT_TICKS = 2000000ull;  // 20 ms: a peer that is later than that is waited for by the host

}  // namespace ndt
