/*
 * ndt_hip.h -- C-ABI of the MI355X-native NDT scan-matching engine.
 *
 * This is the drop-in boundary for the registration path of khalisfadil/slam-sam:
 * the `pcl::Registration<PointXYZI,PointXYZI>::Ptr registration` member of
 * RegisterCallback (ref: include/registercallback.hpp:35) which the drivers fill
 * with a `pclomp::NormalDistributionsTransform` (ref: run/pipeline.cpp:464-481)
 * and drive with setInputTarget / setInputSource / align / getFinalTransformation
 * / getResult (ref: run/pipeline.cpp:557-568, run/pipeline_ligo_tc.cpp:529-538).
 * The C++ adapter include/ndt_hip/ndt_hip.hpp maps those method names onto the
 * functions below; INTEGRATION.md shows the reference-side binding.
 *
 * Conventions (all citations relative to the reference tree):
 *  - plain C: opaque handle, POD structs, caller-allocated outputs; every call
 *    returns an ndt_status (0 = ok, < 0 = error) and never throws.  The engine
 *    needs a gfx950 device: without one every compute entry point fails with
 *    NDT_ERR_NO_DEVICE -- there is no CPU fallback.
 *  - matrices are 4x4 float, COLUMN-major (Eigen::Matrix4f layout), source ->
 *    target frame (ref: run/pipeline.cpp:561,566).
 *  - the 6-vector pose is [x, y, z, roll, pitch, yaw] with R = Rx*Ry*Rz
 *    (ref: extern/svn_ndt/include/svn_ndt_impl.hpp:256-260,272-279).
 *  - score / gradient / Hessian are those of the POSITIVE score that NDT
 *    maximises, so the Hessian is negative definite near the optimum and callers
 *    form cov = -(H + 1e-6 I)^-1 (ref: run/pipeline.cpp:594-596).  Hessians are
 *    6x6 double, row-major (symmetric).
 *  - a handle is NOT thread-safe; distinct handles are independent (the
 *    reference uses one engine per thread, ref: run/pipeline.cpp:432,464).
 *  - the engine never keeps caller pointers after a call returns: a cloud handed
 *    over from host memory has been read completely when ndt_set_target /
 *    ndt_set_source return (the caller may free or overwrite it at once), even
 *    though its transfer and the target's voxel-grid build may still be running
 *    on the device (asynchronous hand-off, below).
 */
#ifndef NDT_HIP_H_
#define NDT_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NDT_HIP_ABI_VERSION 3

typedef enum ndt_status {
  NDT_OK = 0,
  NDT_ERR_INVALID_ARG = -1,
  NDT_ERR_NO_DEVICE = -2,   /* no usable gfx950 device / HIP runtime failure at init */
  NDT_ERR_HIP = -3,         /* a HIP call failed; see ndt_last_error() */
  NDT_ERR_NO_TARGET = -4,   /* align/eval before a target with >= 1 valid voxel */
  NDT_ERR_NO_SOURCE = -5,
  NDT_ERR_GRID_OVERFLOW = -6, /* dx*dy*dz > INT32_MAX (ref: voxel_grid_covariance_impl.hpp:108-125) */
  NDT_ERR_ALLOC = -7,
  NDT_ERR_COMM = -8,        /* RCCL / shared-memory reduction failure */
  NDT_ERR_UNSUPPORTED = -9
} ndt_status;

/* same order as pclomp::NeighborSearchMethod (ref: run/pipeline.cpp:471-480) */
typedef enum ndt_search_method {
  NDT_KDTREE = 0,   /* radius search (radius = resolution) over voxel centroids
                       (ref: voxel_grid_covariance_impl.hpp:505-554), done as a 27-cell scan */
  NDT_DIRECT26 = 1, /* [RECALLED] pclomp getNeighborhoodAtPoint: every valid voxel of the 3x3x3 block around the
                       point's cell, enumerated in integer index space.  The reference tree holds only the
                       enum value (run/pipeline.cpp:471-480) and a commented stub (svn_ndt_impl.hpp:581-583) */
  NDT_DIRECT7 = 2,
  NDT_DIRECT1 = 3
} ndt_search_method;

typedef enum ndt_hessian_mode {
  NDT_HESSIAN_FULL = 0,         /* Magnusson eq. 6.13 (pclomp; svn_ndt_impl.hpp:472-494) */
  NDT_HESSIAN_GAUSS_NEWTON = 1  /* J^T C^-1 J only (svn default, svn_ndt_impl.hpp:496-499) */
} ndt_hessian_mode;

typedef enum ndt_cov_mode {
  NDT_COV_SVN = 0,          /* ss/n - mu mu^T, * n/(n-1)  (voxel_grid_covariance_impl.hpp:287-291) */
  NDT_COV_PCL_RECALLED = 1  /* upstream PCL form, * (n-1)/n -- recalled, unverifiable offline */
} ndt_cov_mode;

typedef enum ndt_reduce_mode {
  NDT_REDUCE_NONE = 0, /* single GPU */
  NDT_REDUCE_RCCL = 1, /* ncclAllReduce of the 32-double partial over xGMI */
  NDT_REDUCE_SHM = 2,  /* pinned-host partials summed through POSIX shared memory */
  NDT_REDUCE_HOOK = 3, /* caller-supplied all-reduce callback */
  NDT_REDUCE_P2P = 4   /* one-shot peer-write all-gather over xGMI + local sum, INSIDE the derivative
                          kernel's final sum: no collective launch, no host hop, and the host-polled /
                          pre-launched fast path of a single GPU stays on */
} ndt_reduce_mode;

typedef enum ndt_wait_mode {
  NDT_WAIT_SPIN = 0,   /* the calling thread polls the result slots in pinned memory (lowest latency;
                          occupies one host core for the duration of ndt_align) */
  NDT_WAIT_BLOCK = 1   /* the calling thread sleeps in hipStreamSynchronize (+3..4 us per evaluation;
                          for hosts that cannot spare a core: the drivers run 6-7 threads + OpenMP) */
} ndt_wait_mode;

typedef enum ndt_source_order {
  NDT_SOURCE_ORDER_AUTO = 0, /* sort when the voxel table is larger than one L2 (records > 6 MB) and the
                                source has >= 32768 points */
  NDT_SOURCE_ORDER_KEEP = 1, /* evaluate the source in the order it was handed over */
  NDT_SOURCE_ORDER_SORT = 2  /* always: once per align, the source is stably sorted by the block of target
                                voxels its points fall into under the initial guess (a wavefront then touches
                                few distinct voxel records; only the f64 summation order changes) */
} ndt_source_order;

typedef enum ndt_prelaunch {
  NDT_PRELAUNCH_AUTO = 0, /* inside ndt_align (spin wait, no RCCL reducer): the kernel of the next evaluation is
                             enqueued while the current one runs and waits on the device for its pose, which the
                             host publishes through BAR-mapped device memory -- takes the launch + dispatch latency
                             (~4 us of ~7) out of every evaluation but the first.  Successive kernels alternate
                             between two streams of the engine, so that the next one takes compute units as the
                             blocks of the one in flight leave.  Used when the device exposes a large BAR; a kernel
                             that waited 20 ms gives up and the pose is evaluated through an ordinary launch.
                             The two-stream placement is right for a device the engine has to itself; AUTO checks that
                             by measurement (the 6th, the 14th and then every 32nd align run with the one-stream placement; one that is 15 %
                             faster per evaluation switches the handle over, and back the same way), so a second
                             engine or process on the same device is noticed without being told. */
  NDT_PRELAUNCH_OFF = 1,
  NDT_PRELAUNCH_ONE_STREAM = 2 /* as AUTO, but every kernel stays on the engine's one stream (the next starts when the
                             current one has ENDED).  For several engines / processes that share ONE device: a
                             waiting kernel holds its compute units, and with two streams it holds them for the
                             whole evaluation of its predecessor -- units the other engine's running kernel needs. */
} ndt_prelaunch;

/* Named parameter sets.  ndt_default_params() is the vendored-code hybrid the parity tests pin
 * (svn covariance, full Hessian, no ridge, More-Thuente); the presets restate the two engines. */
typedef enum ndt_preset {
  NDT_PRESET_DEFAULT = 0,
  NDT_PRESET_PCLOMP_RECALLED = 1, /* upstream pclomp as recalled: PCL covariance normalisation
                                     (n-1)/n, full Hessian, no ridge, More-Thuente line search, min 6 points */
  NDT_PRESET_SVN = 2              /* vendored svn_ndt: n/(n-1) covariance, Gauss-Newton Hessian, +1e-6 I */
} ndt_preset;

/* Parameter block; mirrors the pclomp / svn_ndt setters the drivers call and
 * RegisterCallback's JSON fields (ref: include/registercallback.hpp:37-54,
 * src/registercallback.cpp:24-91). */
typedef struct ndt_params {
  float resolution;              /* setResolution; voxel leaf size in metres */
  double step_size;              /* setStepSize; More-Thuente step_max */
  double trans_epsilon;          /* setTransformationEpsilon */
  int max_iterations;            /* setMaximumIterations */
  double outlier_ratio;          /* setOutlierRatio (0.55) */
  int search_method;             /* ndt_search_method */
  int min_points_per_voxel;      /* 6; clamped to >= 3 (voxel_grid_covariance.h:153,176-184) */
  double eig_inflation_ratio;    /* 0.01 (voxel_grid_covariance.h:154) */
  int hessian_mode;              /* ndt_hessian_mode */
  int cov_mode;                  /* ndt_cov_mode */
  int add_ridge;                 /* H += 1e-6 I after accumulation (svn_ndt_impl.hpp:650-653) */
  int use_line_search;           /* 1: More-Thuente; 0: fixed step min(|dp|, step_size) */
  float regularization_scale_factor; /* setRegularizationScaleFactor (run/pipeline_ligo_tc.cpp:293) */
  int num_threads;               /* setNumThreads; recorded only -- there is no CPU path */
  int device_id;                 /* HIP device ordinal; -1 = current device */
  int wait_mode;                 /* ndt_wait_mode */
  int source_order;              /* ndt_source_order */
  int prelaunch;                 /* ndt_prelaunch */
} ndt_params;

typedef struct ndt_handle ndt_handle;

/* pclomp::NdtResult + pcl::Registration outputs (ref: run/pipeline.cpp:566-568,594;
 * include/map.hpp:96-97 for the two timing/iteration statistics). */
typedef struct ndt_result {
  float final_transformation[16]; /* getFinalTransformation(), column-major */
  double final_pose[6];           /* [x,y,z,roll,pitch,yaw] */
  int converged;                  /* hasConverged() */
  int iterations;                 /* getFinalNumIteration() / NdtResult::iteration_num */
  int n_evaluations;              /* derivative evaluations incl. line-search trials */
  double hessian[36];             /* NdtResult::hessian, row-major */
  double score;
  double transform_probability;   /* score / #source points */
  double nearest_voxel_transformation_likelihood;
  int64_t n_pairs;                /* (point, voxel) pairs of the last evaluation */
  int64_t n_points_with_neighbors;
  double ms_total;                /* wall time of ndt_align */
  double ms_device;               /* sum of device-side evaluation time (HIP events) */
  int n_evaluations_reused;       /* line-search requests at the pose of the evaluation before them
                                     (a step clamped to its lower bound is re-tried up to 10 times):
                                     answered without a launch, bit-identical by construction */
} ndt_result;

/* Per-voxel statistics, the accessors extractNdtData() uses
 * (ref: include/pipeline.hpp:175-206; Leaf: voxel_grid_covariance.h:99-131). */
typedef struct ndt_leaf {
  int64_t index;     /* 1-D voxel index ijk0 + ijk1*div_x + ijk2*div_x*div_y */
  int32_t point_count;
  float center[3];   /* getLeafCenter(index) */
  double mean[3];
  double cov[9];     /* row-major */
  double icov[9];
  double evecs[9];   /* eigenvectors as columns, eigenvalues ascending */
  double evals[3];
} ndt_leaf;

typedef struct ndt_grid_info {
  int min_b[3], max_b[3], div_b[3];
  float leaf_size, inverse_leaf_size;
  int64_t n_leaves;        /* valid voxels */
  int64_t n_cells;         /* dx*dy*dz of the dense index grid */
  int64_t n_target_points;
  double ms_build;         /* time of the last target build: device time (HIP events) while kernel timing is enabled, wall time of the call otherwise */
} ndt_grid_info;

/* one derivative evaluation: [score, g(6), H upper-tri (21), nvtl_sum,
 * n_points_with_neighbors, n_pairs, pad] */
#define NDT_EVAL_WORDS 32

/* ---- lifecycle ---------------------------------------------------------- */
int ndt_abi_version(void);
void ndt_default_params(ndt_params* p);
/* overwrites the algorithm switches of *p (cov_mode, hessian_mode, add_ridge, use_line_search,
 * min_points_per_voxel) with a named set; everything else is left as it is */
int ndt_params_preset(ndt_params* p, int preset /* ndt_preset */);
int ndt_create(const ndt_params* p, ndt_handle** out);   /* new pclomp::NormalDistributionsTransform */
int ndt_destroy(ndt_handle* h);
int ndt_set_params(ndt_handle* h, const ndt_params* p);  /* a changed resolution rebuilds the grid */
int ndt_get_params(const ndt_handle* h, ndt_params* p);
const char* ndt_last_error(const ndt_handle* h);
/* human-readable device description; returns the number of visible devices or < 0 */
int ndt_backend_info(char* buf, size_t cap);

/* ---- clouds ------------------------------------------------------------- */
/* Hand-off of HOST clouds (the drivers hold host pcl::PointCloud<PointXYZI>, ref: run/pipeline.cpp:554-561).
 * NDT_HANDOFF_ASYNC (default): ndt_set_target / ndt_set_source (and their _soa forms) return as soon as the
 * caller's memory has been repacked into the engine's pinned staging; the PCIe copies, the SoA conversion and the
 * target's voxel-grid build run behind on the device, the target's under the source's repack, and the first call
 * that needs them (ndt_align, ndt_eval_derivatives, ndt_get_grid_info, ndt_export_leaves, ...) waits for them.
 * Consequence: a steady-state build that FAILS (no finite point, index overflow) is reported by that first call
 * (or by ndt_wait), with the same status code, not by ndt_set_target -- the reference's setInputTarget returns
 * void, its failures surface in align the same way (ref: svn_ndt_impl.hpp:682-702).  The first build of a handle,
 * the build that follows a failed one (both wait for the grid geometry inside the call), an empty cloud and
 * argument errors are still reported at once.
 * NDT_HANDOFF_SYNC: both calls block until the device has everything (rounds 1-3 behaviour; also NDT_HANDOFF=sync
 * in the environment). */
typedef enum ndt_handoff_mode { NDT_HANDOFF_ASYNC = 0, NDT_HANDOFF_SYNC = 1 } ndt_handoff_mode;
/* Tuning and A/B switches (round 5).  The production library reads NONE of these from the environment -- a library
 * linked into somebody else's process is not steered by variables that process never heard of.  The environment is
 * left with four documented OPERATIONAL knobs: NDT_HANDOFF=sync (blocking hand-off), NDT_UPLOAD_THREADS (repack
 * workers), NDT_COMM_TIMEOUT_S (multi-rank reduce time-out), NDT_PRELAUNCH=0 (no pre-launched kernels).  Everything
 * else is this struct: process-wide, read at the launches and handle creations that FOLLOW the call; every field has
 * the default ndt_get_tuning() reports in a fresh process.  Results do not depend on any of them (same rows, same
 * order, same bits), only timings do -- except deriv_block / deriv_single_level_max, which change the partition of
 * the scan and with it the last bits of the floating-point sums.
 * (The diagnostic library variants -- make VARIANT=ab|seams|stamps -- still take the historical NDT_* variables as
 * initial values of these fields; the tuning programs under tools/ use ndt_set_tuning.) */
typedef struct ndt_tuning {
  int deriv_block;            /* 0: chosen per launch (default); else threads per block of k_derivatives, multiple of 64, 64..1024 */
  int deriv_summer;           /* 1: a fixed block adds the partial rows by polling their tags (default); 0: ticket, last block adds */
  int deriv_dedicated;        /* 1: that block owns no points where a compute unit is spare (default); 0: block 0 doubles */
  int deriv_single_level_max; /* rows one block adds directly (default 2048); larger grids go through 32 group rows */
  int deriv_xcd;              /* 0: chunk = block id; 1: XCD-aware chunks on resident single-pose grids (default); 2: stripes too */
  int bucket_build;           /* 1: steady-state builds in two launches (default); 0: launch-per-phase sort pipeline */
  int bucket_tile;            /* 0: chosen from the cloud size (default); 1024 | 2048 | 4096 | 8192 points per tile of k_bucket_pass */
  int fused_sort;             /* 1: one launch per sort digit where the cloud allows it (default); 0: classic passes */
  int bounds_blocks;          /* blocks of the bounds pass (default 256) */
  int bounds_unroll;          /* 8 (default) | 4 */
  int finalize_threads;       /* 256 (default) | 64 */
  int build_events;           /* -1: HIP events around a build only while kernel timing is on (default); 0 | 1: never | always */
  int build_wait_sync;        /* 0: the host polls the build's done tag (default); 1: hipStreamSynchronize */
  int mbox_tagged;            /* 1: poses reach pre-launched kernels as tagged 8-byte granules (default); 0: words + sequence */
  int mbox_preload;           /* 0 (default); 1: a pre-launched kernel fetches its point before the pose arrives */
  int prelaunch_streams;      /* 2: pre-launched kernels on the engine's second stream (default); 1: one stream */
  int prelaunch_probe;        /* 1: the automatic stream placement probes the other placement (default); 0: never */
  int speculate_first;        /* 1: first evaluation of an align enqueued behind a running build (default); 0: off */
  int timing_bracket;         /* 0: kernel-timing events attached to the dispatch (default); 1: recorded around the launch call */
  int handoff_chunk_pass;     /* 0: the asynchronous host hand-off partitions the target behind its transfer (default); 1: under it, chunk by
                               * chunk on a stream of its own (measured slower: profiles/r05_handoff_chunk_pass_ab.txt) */
  int deriv_summer_split;     /* 1: four summing blocks, one 128-byte line of every row each, where compute units are spare (default); 0: one;
                               * 4 | 8: that many (A/B) */
  int deriv_one_block_per_cu; /* 1: a single-pose launch of at most one block per compute unit asks for more than half a unit's LDS, so that no
                               * two of its blocks share a unit (default); 0: blocks of <= 8 waves may */
  int reserved[10];           /* zero */
} ndt_tuning;
/* Idle-device heartbeat (round 5; default off).  A driver at the reference's 10-20 Hz keyframe rate leaves the device idle for
 * 50-100 ms between two aligns, and an idle MI355X drops its clocks: the align that follows runs 5-10 % slower than in a
 * busy loop.  period_us > 0 (>= 100): while the handle has been idle for a period, a background thread
 * launches one small kernel (one block per compute unit, ~3 us) per period on a lowest-priority stream; 0 stops it.
 * ndt_get_keepwarm returns the period (0: off) and, if beats != NULL, the number of beats launched so far. */
int ndt_set_keepwarm(ndt_handle* h, int period_us);
int ndt_get_keepwarm(const ndt_handle* h, long long* beats);

int ndt_get_tuning(ndt_tuning* out);
/* NDT_ERR_INVALID_ARG (nothing changed) when a field is outside its documented values. */
int ndt_set_tuning(const ndt_tuning* t);

int ndt_set_handoff_mode(ndt_handle* h, int mode);
int ndt_get_handoff_mode(const ndt_handle* h);
/* Blocks until every hand-off in flight is complete on the device; returns the status of a deferred build that
 * failed, NDT_OK otherwise.  A deferred failure is reported once -- by this call or by the first call that needs the
 * grid, whichever comes first; after that the handle is where a failed blocking ndt_set_target leaves it (no target:
 * NDT_ERR_NO_TARGET from the calls that need one). */
int ndt_wait(ndt_handle* h);

/* setInputTarget (ref: run/pipeline.cpp:557): uploads and builds the voxel grid.
 * xyz points to the first x; consecutive points are stride_bytes apart (12 for
 * packed xyz, 32 for pcl::PointXYZI, 16 for pcl::PointXYZ). */
int ndt_set_target(ndt_handle* h, const float* xyz, size_t n, size_t stride_bytes);
/* SoA host arrays (LidarFrame x/y/z, ref: include/dataframe.hpp:344-346) */
int ndt_set_target_soa(ndt_handle* h, const float* x, const float* y, const float* z, size_t n);
/* SoA arrays already resident in device memory (consumed during the call) */
int ndt_set_target_device(ndt_handle* h, const float* dx, const float* dy, const float* dz, size_t n);
/* As ndt_set_target_device, but under NDT_HANDOFF_ASYNC a steady-state build is only ENQUEUED on the engine's stream:
 * the arrays must stay valid and unchanged until the first call that needs the grid (ndt_align, ndt_get_grid_info,
 * ndt_wait, ...) has returned, and a failed build is reported by that call.  What it buys: the align that follows
 * enqueues its first derivative evaluation BEHIND the build while it still runs (the kernel reads the grid geometry
 * from device memory), so the launch is not paid after the build's verdict -- as for every deferred build: host
 * clouds under NDT_HANDOFF_ASYNC, ndt_set_target_from_keyframes. */
int ndt_set_target_device_deferred(ndt_handle* h, const float* dx, const float* dy, const float* dz, size_t n);
/* setInputSource (ref: run/pipeline.cpp:558) */
int ndt_set_source(ndt_handle* h, const float* xyz, size_t n, size_t stride_bytes);
int ndt_set_source_soa(ndt_handle* h, const float* x, const float* y, const float* z, size_t n);
int ndt_set_source_device(ndt_handle* h, const float* dx, const float* dy, const float* dz, size_t n);
/* The same without the copy: the engine reads the caller's device arrays at every later evaluation,
 * so they must stay valid and unchanged until the source is replaced -- pcl::Registration::
 * setInputSource's contract (it keeps the caller's shared_ptr; ref: run/pipeline.cpp:558). */
int ndt_set_source_device_view(ndt_handle* h, const float* dx, const float* dy, const float* dz, size_t n);
/* The engine may cache a block-ordered COPY of a viewed source (ndt_source_order).  After rewriting
 * the viewed arrays in place (a reused scan buffer) call ndt_set_source_device_view again, or this
 * notice (no copy, no launch): later aligns then re-derive whatever was cached from the arrays.
 * Freeing the arrays while they are the source is the caller's error, as with the shared_ptr. */
int ndt_source_changed(ndt_handle* h);

/* Voxel-record format of the derivative kernel (SURVEY 7 "packed 48-B record").  NDT_RECORDS_F64: 80 bytes per
 * voxel, mean and inverse covariance in f64 (default; what the 1e-9 parity tests run on).  NDT_RECORDS_PACKED48:
 * 48 bytes, the mean stays f64 and the inverse covariance is rounded to f32 -- what the reference's own per-pair
 * gradient / Hessian arithmetic does with it (c_inv4, svn_ndt_impl.hpp:449-456); only the score's Mahalanobis term
 * sees an f32 matrix where the reference keeps f64.  Derivatives move by ~1e-7 of their norm, the aligned transform
 * by micrometres (tests/test_gpu_features.py); an evaluation fetches three 16-byte pieces per neighbour instead of
 * five.  Applies to the DIRECT7 / DIRECT1 neighbourhoods; KDTREE, DIRECT26 and a multi-grid union keep reading the
 * 80-byte records (measured: no gain there; the union chains its leaves through them).  Takes effect at
 * the next evaluation; exported leaf statistics are the f64 ones either way. */
typedef enum ndt_record_format { NDT_RECORDS_F64 = 0, NDT_RECORDS_PACKED48 = 1 } ndt_record_format;
int ndt_set_record_format(ndt_handle* h, int format);
int ndt_get_record_format(const ndt_handle* h);

/* Multi-grid target [RECALLED: tier4 ndt_omp's MultiGridNormalDistributionsTransform -- addTarget /
 * removeTarget / createVoxelKdtree -- named by the reference's build (CMakeLists.txt:41-42); its
 * sources are in the absent extern/ndt_omp submodule and no driver instantiates it].  Every cloud is
 * voxelised on its own (the leaves of setInputTarget on that cloud alone) and kept under an id;
 * ndt_multigrid_create_kdtree makes the union of all stored grids the target: the neighbourhood is
 * the radius search (radius = leaf size) over the centroids of ALL grids' valid leaves, whatever
 * search_method says; a voxel that two grids share contributes once per grid.  Afterwards
 * ndt_align / ndt_score_transform / ndt_eval_derivatives work as after ndt_set_target; a plain
 * ndt_set_target* call replaces the union (the stored grids stay until removed). */
int ndt_multigrid_add_target(ndt_handle* h, int64_t id, const float* xyz, size_t n, size_t stride_bytes);
int ndt_multigrid_remove_target(ndt_handle* h, int64_t id);
int64_t ndt_multigrid_count(const ndt_handle* h);
int ndt_multigrid_create_kdtree(ndt_handle* h);

/* Device-resident keyframe archive + sliding-window target assembly.  The drivers keep every
 * keyframe's body-frame scan (pointsArchive, ref: run/pipeline.cpp:784) and rebuild the NDT
 * target per keyframe as the sum of <= 5 archived scans, each moved by its current pose
 * (ref: run/pipeline_ligo_tc.cpp:519-529; one scan in run/pipeline.cpp:554-557).  Here the
 * scans stay in HBM; only ids and 4x4 double poses cross the boundary per keyframe. */
int ndt_keyframe_put(ndt_handle* h, int64_t id, const float* xyz, size_t n, size_t stride_bytes);
int ndt_keyframe_erase(ndt_handle* h, int64_t id);
int64_t ndt_keyframe_count(const ndt_handle* h);
/* target = concat_k transform(archive[ids[k]], poses16[k]) (f64 transform, rounded to f32 once,
 * as pcl::transformPointCloud with a double matrix), then the voxel-grid build */
int ndt_set_target_from_keyframes(ndt_handle* h, const int64_t* ids, const double* poses16,
                                  int n_keyframes);
/* setInputSource(archive[id]): the scan that was just archived is also the one to register
 * (ref: run/pipeline.cpp:558 registers pointsBody, :784 archives the same cloud) -- one upload
 * serves both.  The source VIEWS the archived scan (no copy): erasing or replacing that keyframe
 * unsets the source (NDT_ERR_NO_SOURCE until one is set again). */
int ndt_set_source_from_keyframe(ndt_handle* h, int64_t id);

/* pcl::VoxelGrid downsample on the device (ref: run/pipeline_ins_map_distribution.cpp:324-340: the accumulated map
 * is filtered at `mapvoxelsize` before the NDT export).  PCL's published algorithm: the grid of
 * getMinMax3D over the finite points, voxel index floor(p * inv_leaf) - min_b, every field averaged (in float) over
 * the points of an occupied voxel, output in ascending voxel index; non-finite points are dropped; a grid of more than
 * INT32_MAX cells is refused (NDT_ERR_GRID_OVERFLOW).  Within a voxel the points are added in input order.
 *  - _device: SoA float arrays in device memory in and out (intensity arrays may be NULL); at most `cap` points are
 *    written, *n_out receives the number of occupied voxels (NDT_ERR_INVALID_ARG if it exceeds cap; cap = n always
 *    suffices).  The output arrays can go straight into ndt_set_target_device: no host round trip.
 *  - host form: a strided cloud in (stride_bytes apart; intensity at intensity_offset_bytes, < 0: none --
 *    pcl::PointXYZI: stride 32, intensity at 16) and a cloud of the same layout out (only x, y, z and intensity are
 *    written).
 * The handle's target grid and source are left untouched. */
int ndt_voxel_downsample_device(ndt_handle* h, const float* dx, const float* dy, const float* dz, const float* d_intensity,
                                size_t n, float leaf, float* ox, float* oy, float* oz, float* o_intensity, size_t cap,
                                size_t* n_out);
int ndt_voxel_downsample(ndt_handle* h, const float* xyz, size_t n, size_t stride_bytes, long intensity_offset_bytes,
                         float leaf, float* out, size_t cap, size_t* n_out);

/* setRegularizationPose (ref: run/pipeline_ligo_tc.cpp:531) */
int ndt_set_regularization_pose(ndt_handle* h, const float pose_colmajor[16]);
int ndt_clear_regularization_pose(ndt_handle* h);

/* ---- registration ------------------------------------------------------- */
/* align(out, guess) / computeTransformation (ref: run/pipeline.cpp:561,
 * test_svn_ndt.cpp:171).  Blocks until the result is on the host. */
int ndt_align(ndt_handle* h, const float guess_colmajor[16], ndt_result* out);

/* pclomp::NdtResult's per-iteration arrays [RECALLED: tier4 ndt_omp's transformation_array,
 * transform_probability_array, nearest_voxel_transformation_likelihood_array; the reference's drivers read none of them]
 * of the LAST ndt_align on this handle: entry 0 is the initial guess with the scores of the first evaluation, then one
 * entry per Newton iteration (the transform after it, the scores of its accepted evaluation).  Any output may be NULL;
 * at most `cap` entries are written; returns the number of entries available (iterations + 1) or < 0. */
int ndt_get_iteration_history(const ndt_handle* h, float* transforms16_colmajor, double* transform_probability,
                              double* nearest_voxel_transformation_likelihood, int cap);

/* Derivatives at K poses in one call (computeDerivatives; and Stage 1 of
 * svn_ndt::align, ref: svn_ndt_impl.hpp:758-781).  poses6: K x 6 doubles.
 * transforms: optional K x 16 floats (column-major) applied to the source;
 * NULL = the matrix built from each pose.  out: K x NDT_EVAL_WORDS doubles. */
int ndt_eval_derivatives(ndt_handle* h, const double* poses6, const float* transforms,
                         int K, int compute_hessian, double* out);
/* unpack one evaluation into score, g[6], H[36] (row-major) */
void ndt_unpack_eval(const double* eval_words, double* score, double* g6, double* H36);
/* The two pose-only ingredients of an evaluation as the engine computes them on the host, no handle and no device
 * needed (round 5; the C++ adapter builds the reference's public per-pair math hooks on them):
 * - the angular tables of computeAngleDerivatives for pose [x, y, z, roll, pitch, yaw] (ref: svn_ndt_impl.hpp:254-331):
 *   j_ang as 8 rows x 3 floats, h_ang as 15 rows x 3 floats -- the words every k_derivatives launch receives;
 * - the Gaussian constants d1, d2 of updateNdtConstants (ref: svn_ndt_impl.hpp:90-130). */
int ndt_angle_tables(const double pose6[6], float j_ang[24], float h_ang[45]);
int ndt_gauss_constants(double resolution, double outlier_ratio, double* d1, double* d2);

/* Scoring-only evaluation (pclomp's calculateTransformationProbability /
 * calculateNearestVoxelTransformationLikelihood [RECALLED]; the reference names them only through
 * SURVEY 8f-4): score, transform probability (score / #source points) and NVTL (mean over the points
 * that have a neighbour of their best per-voxel score) of the current source under T against the
 * current target.  One launch of the score-only kernel: no gradient, no Hessian. */
typedef struct ndt_score {
  double score;
  double transform_probability;
  double nearest_voxel_transformation_likelihood;
  int64_t n_pairs;
  int64_t n_points_with_neighbors;
} ndt_score;
int ndt_score_transform(ndt_handle* h, const float T_colmajor[16], ndt_score* out);
/* K transforms (K x 16 floats) scored in ONE launch of the batched score-only kernel */
int ndt_score_transforms(ndt_handle* h, const float* transforms_colmajor, int K, ndt_score* out);

/* 2-D (x, y) covariance estimators of tier4 ndt_omp's estimate_covariance.cpp.
 * [RECALLED]: the file is in the un-vendored submodule; the reference names it only in its build
 * (ref: CMakeLists.txt:40) and no driver calls it.  cov_xy: 2x2 row-major.
 *  - Laplace approximation: -(H[0:2,0:2])^-1 of a result's Hessian;
 *  - poses to search: (offset_x, offset_y) pairs rotated onto the principal axes of that
 *    covariance and added to the result's translation (n x 16 floats out);
 *  - MULTI_NDT: re-align from every pose, unbiased sample covariance of the (x, y) of the main
 *    result and the n re-aligned results (the handle's last result is the last re-alignment);
 *  - MULTI_NDT_SCORE: NVTL of the source at every pose -- one batched launch --, weights
 *    softmax(NVTL / temperature) over {main result} + poses, weighted mean and covariance. */
int ndt_xy_covariance_laplace(const double hessian36[36], double cov_xy[4]);
int ndt_propose_poses_to_search(const ndt_result* r, const double* offsets_x, const double* offsets_y, int n,
                                float* poses16_out);
int ndt_xy_covariance_multi_ndt(ndt_handle* h, const ndt_result* main_result, const float* poses16, int n,
                                double mean_xy[2], double cov_xy[4]);
int ndt_xy_covariance_multi_ndt_score(ndt_handle* h, const ndt_result* main_result, const float* poses16, int n,
                                      double temperature, double mean_xy[2], double cov_xy[4]);

/* Covariance of a registration result for the pose graph: cov = -(H + eps I)^-1 of
 * ndt_result.hessian (ref: run/pipeline.cpp:594-596, eps = 1e-6 there), and with
 * gtsam_order != 0 the block permutation of RegisterCallback::reorderCovarianceForGTSAM
 * (ref: src/registercallback.cpp:170-186: rotation block first, cross blocks left where
 * they are).  Host-only, no handle.  NDT_ERR_INVALID_ARG when H + eps I is singular or
 * not finite. */
int ndt_result_covariance(const double hessian36[36], double eps, int gtsam_order, double cov36[36]);

/* output cloud of align(): source transformed by T (device-side), packed xyz */
int ndt_transform_source(ndt_handle* h, const float T_colmajor[16], float* out_xyz, size_t cap_points);

/* ---- voxel grid accessors ------------------------------------------------ */
int ndt_get_grid_info(const ndt_handle* h, ndt_grid_info* out);
/* getTargetCells().getLeaves(): valid leaves sorted by ascending index; returns
 * the number written (<= cap) or < 0 */
int64_t ndt_export_leaves(ndt_handle* h, ndt_leaf* out, size_t cap);

/* ---- host Newton driver with an external evaluator ----------------------- */
/* fn must fill out[NDT_EVAL_WORDS] with the GLOBAL (already reduced) evaluation
 * at pose6 / T; return 0 on success. */
typedef int (*ndt_eval_fn)(void* ctx, const double pose6[6], const float T_colmajor[16],
                           int compute_hessian, double out[NDT_EVAL_WORDS]);
int ndt_newton_align(const ndt_params* p, int64_t n_source_total,
                     const float guess_colmajor[16], const float* regularization_pose_or_null,
                     ndt_eval_fn fn, void* ctx, ndt_result* out);

/* ---- multi-GPU: one process per GPU, source sharded, target replicated ---- */
/* contiguous shard of n items for rank r of nranks */
void ndt_shard_range(size_t n, int rank, int nranks, size_t* begin, size_t* count);
/* rank 0 creates the id (128 bytes) and the caller broadcasts it (e.g. through
 * torch.distributed); then every rank calls ndt_comm_init_rccl. */
int ndt_comm_unique_id(void* out128);
int ndt_comm_init_rccl(ndt_handle* h, const void* id128, int rank, int nranks);
/* host-side reduction through a POSIX shared-memory segment named `name` */
int ndt_comm_init_shm(ndt_handle* h, const char* name, int rank, int nranks);
/* Peer-write reducer (NDT_REDUCE_P2P).  Every rank calls ndt_comm_p2p_handle (allocates its exchange
 * area in fine-grained device memory on the handle's device and exports it as an IPC handle of
 * NDT_P2P_HANDLE_BYTES), the caller all-gathers the handles (bench.py: over its shared-memory board), then
 * every rank calls ndt_comm_init_p2p with all of them in rank order.  Ranks are processes; all their
 * devices must be visible to each other (no HIP_VISIBLE_DEVICES masking per rank). */
#define NDT_P2P_HANDLE_BYTES 64
int ndt_comm_p2p_handle(ndt_handle* h, void* out_handle);
int ndt_comm_init_p2p(ndt_handle* h, const void* handles, int rank, int nranks);
/* First-contact instrumentation of the peer-write reducer (round 5).
 * ndt_comm_p2p_selftest: COLLECTIVE (every rank, same `rounds`; put a barrier of your own behind it): `rounds` lock-step rounds
 * of patterned {tag, value} slots through the exchange areas, written and read exactly as the derivative kernel's final sum
 * does it -- the direct test of "a 16-byte slot is seen entirely old or entirely new" across devices.  out: {rounds completed,
 * slots seen with a new tag and an old value, rounds a peer missed (the pass stops at the first), longest round in 10 ns ticks}.
 * ndt_comm_p2p_stats: out = {exchanges made inside a kernel's final sum since ndt_comm_init_p2p (or the last reset), their
 * summed duration and the longest one in 10 ns ticks (own row published -> every rank's row read), exchanges a peer was late for}. */
int ndt_comm_p2p_selftest(ndt_handle* h, int rounds, int64_t out[4]);
int ndt_comm_p2p_stats(ndt_handle* h, int64_t out[4], int reset);
/* caller-supplied all-reduce(sum) over NDT_EVAL_WORDS doubles, in place */
typedef int (*ndt_allreduce_fn)(void* ctx, double* words, int n);
int ndt_comm_init_hook(ndt_handle* h, ndt_allreduce_fn fn, void* ctx, int rank, int nranks);
int ndt_comm_destroy(ndt_handle* h);
/* which collective library serves the RCCL reducer: ncclGetVersion() code (e.g. 22105) and the
 * path of the shared object the symbol was resolved from (the process may hold two librccl.so:
 * ROCm's and the one bundled with PyTorch); returns the version or < 0 */
int ndt_comm_info(char* path_buf, size_t cap);
/* ranks of the handle's live reducer as the transport reports them (RCCL: ncclCommCount of the engine's
 * communicator); 1 without a reducer; < 0 on failure */
int ndt_comm_rank_count(const ndt_handle* h);
/* total source points over all ranks (for transform_probability); set by the
 * caller after sharding, defaults to the local count */
int ndt_set_global_source_size(ndt_handle* h, int64_t n_total);

/* ---- SVN-NDT (svn_ndt::SvnNormalDistributionsTransform::align) ------------- */
/* Stein Variational Newton over K pose particles (ref: extern/svn_ndt/include/svn_ndt.h:
 * 100-182, svn_ndt_impl.hpp:675-964; driver: run/pipeline_lo_svn.cpp:301-319,387-388).
 * Stage 1 (NDT derivatives of every particle) is one batched kernel launch; stages 2-3
 * run on the host.  The engine's ndt_params select the svn defaults through
 * hessian_mode = NDT_HESSIAN_GAUSS_NEWTON and add_ridge = 1 (svn_ndt.h:314,
 * svn_ndt_impl.hpp:650-653).  Poses are 4x4 double, column-major; covariance 6x6 row-major
 * in GTSAM tangent order [rot, trans] (svn_ndt.h:46). */
typedef struct ndt_svn_params {
  int particle_count;      /* setParticleCount (30) */
  int max_iterations;      /* setMaxIterations (50) */
  double kernel_bandwidth; /* setKernelBandwidth (1.0) */
  double step_size;        /* setStepSize (1.0) */
  double stop_threshold;   /* setEarlyStopThreshold (1e-4) */
} ndt_svn_params;

typedef struct ndt_svn_result {  /* svn_ndt::SvnNdtResult (svn_ndt.h:40-51) */
  double final_pose[16];
  double final_covariance[36];
  int converged;
  int iterations;
  double last_mean_update;   /* |Log(mean_prev^-1 mean)| of the last iteration */
  double ms_total, ms_stage1, ms_stage2, ms_stage3;
} ndt_svn_result;

void ndt_svn_default_params(ndt_svn_params* p);
/* prior.retract(sigma * N(0,1)) for K particles (ref :708-716); the reference seeds from the
 * wall clock, here the seed is explicit.  particles16: K x 16 doubles. */
int ndt_svn_sample_particles(const double prior16[16], int K, uint64_t seed, double* particles16);
/* The particle kernel of Stage 2 for ONE pair, as ndt_svn_align evaluates it (round 5; host arithmetic, no handle):
 * k = exp(-|Log(l^-1 k)|^2 / bandwidth) and, if grad6 != NULL, its gradient with respect to l in l's tangent space,
 * [rotation, translation] (ref: rbf_kernel / rbf_kernel_gradient, svn_ndt_impl.hpp:213-244).  Poses: 4 x 4 column-major. */
int ndt_svn_rbf_kernel(const double pose_l16[16], const double pose_k16[16], double bandwidth, double* k, double* grad6);
/* particles16 (K x 16): initial particles in, final particles out.  The source / target
 * clouds are those of the handle (ndt_set_target / ndt_set_source). */
int ndt_svn_align(ndt_handle* h, const ndt_svn_params* p, const double prior16[16],
                  double* particles16, ndt_svn_result* out);

/* ---- instrumentation ------------------------------------------------------ */
typedef struct ndt_timing {
  double ms_last_eval_kernel;   /* HIP-event time of the last derivative kernel */
  double ms_last_reduce_kernel; /* round 5: wall time of the last CROSS-RANK sum on the host (shm / hook transports; 0 without
                                 * a reducer; the peer-write exchange runs inside the kernel: ndt_comm_p2p_stats) */
  double ms_last_build;
  int64_t n_eval_launches;      /* since handle creation */
  double ms_eval_kernel_total;  /* summed HIP-event time of the accumulation kernel while timing is on */
  double ms_reduce_kernel_total;
  int64_t n_timed_evals;        /* evaluations covered by the two totals */
} ndt_timing;
int ndt_enable_kernel_timing(ndt_handle* h, int on);
int ndt_get_timing(const ndt_handle* h, ndt_timing* out);

/* Breakdown of the last host hand-off of each cloud (ndt_set_target* / ndt_set_source* from host memory). */
typedef struct ndt_handoff_lane_timing {
  int64_t n_points;
  int64_t bytes_in;     /* bytes of the caller's cloud that were read (n x stride, or 12 n for SoA) */
  int64_t bytes_dma;    /* bytes that crossed PCIe (12 n) */
  double ms_repack;     /* host time of the call up to its return in asynchronous mode: wait for the staging buffer,
                           AoS -> chunk-major repack on `threads` threads, issue of the copies and the SoA kernel */
  double ms_dma;        /* device time from before the first copy to behind the last, by HIP events; only while kernel
                           timing is enabled (ndt_enable_kernel_timing) and once the hand-off has completed */
  double dma_gb_per_s;  /* bytes_dma / ms_dma */
  int threads;          /* repack threads incl. the caller */
} ndt_handoff_lane_timing;
typedef struct ndt_handoff_timing {
  ndt_handoff_lane_timing target, source;
  double ms_build_wait; /* time the first call that needed the grid waited for the deferred build's verdict */
  int mode;             /* ndt_handoff_mode */
  int cpu_budget;       /* CPUs this process may use (affinity mask cut down to the cgroup quota) */
  int repack_workers;   /* worker threads of the repack pool (NDT_UPLOAD_THREADS overrides) */
} ndt_handoff_timing;
int ndt_get_handoff_timing(const ndt_handle* h, ndt_handoff_timing* out);

#ifdef __cplusplus
}
#endif
#endif /* NDT_HIP_H_ */
