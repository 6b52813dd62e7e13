// ndt_handoff.hip -- host hand-off, orchestration of the target voxel-grid build, grid accessors (see ndt_engine.h).
#include "ndt_engine.h"

namespace ndt {
namespace engine {

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
               DevBuf<float>& dz, bool sync, const std::function<void(size_t, bool)>* after_chunk) {
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
    if (after_chunk) (*after_chunk)(hi, hi >= n);
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
  HIP_TRY(h, h->bucket_tab.ensure(bucket_table_words()));
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
  // the two-launch build: steady state only (decided per attempt below)
  br.bucketed_ok = bucket_build_enabled() && bucket_build_fits(n, h->n_cus);
  // A cloud the two-launch build declined (BG_BUCKET: a bucket beyond a block's LDS or hash table, far-away coordinates)
  // is usually followed by more of its kind (the same map, the next keyframe): the attempt costs two launches and, for a
  // late decline, a full clear of the index grid, so after a decline the next 8 builds go sort-based straight away, 16 after
  // the next decline, ... at most 64.
  if (br.bucketed_ok && h->bucket_skip > 0) {
    --h->bucket_skip;
    br.bucketed_ok = false;
  }
  if (br.fused_sort && !h->sort_tags.p) {
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
    if (br.pass_chunked && br.attempt == 0 && br.pass_tiles == bucket_build_tiles(n)) {
      // the partition ran under the transfer (chunked_pass_hook): only the leaves launch is left, and the build's clock
      // starts here
      br.t0 = std::chrono::steady_clock::now();
      if (br.build_events) HIP_TRY(h, hipEventRecord(h->ev0, s));
    } else {
      HIP_TRY(h, launch_bucket_pass_tiles(x, y, z, n, br.inv_leaf, h->bucket_tab.p, h->stats.p, br.dirty_slots, h->cell2leaf.p,
                                          h->cell2leaf.cap, h->bnd.p, h->nleaf.p, h->xyz4.p, 0, bucket_build_tiles(n), s));
    }
    HIP_TRY(h, launch_bucket_leaves(n, br.leaf, br.inv_leaf, cap_cells, min_pts, fpb, h->gd.p, h->gdh.d, h->bucket_tab.p,
                                    h->cell2leaf.p, h->bnd.p, h->nleaf.p, h->tickets.p + 4, h->xyz4.p, h->leaf_sums.p, h->rec.p,
                                    h->cent.p, h->stats.p, max_leaves, h->small.d + 8, done_tag, s));
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
  if (br.bucketed && bg.status == BG_BUCKET) {
    // declined: huge coordinates (decided before anything was written), or a bucket beyond a block's LDS or hash table --
    // found by that bucket's block after others had published leaves.  Once more, sort-based; the retry does not trust
    // the grid: full clear, geometry awaited.  (Neither launch waits for a sibling block since round 5: no BG_SPIN here.)
    br.bucketed_ok = false;
    br.dirty_slots = 0;
    br.clean_cap = 0;
    h->bucket_backoff = std::min(64, std::max(8, 2 * h->bucket_backoff));
    h->bucket_skip = h->bucket_backoff;
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
int build_grid(ndt_handle* h, const float* x, const float* y, const float* z, size_t n, bool defer) {
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

// The asynchronous hand-off of a HOST target.  ndt_tuning::handoff_chunk_pass = 1 (OFF by default): the
// two-launch build's partition needs no grid geometry and no sibling tile (round 5), so it can run UNDER the transfer --
// behind every chunk's pull kernel the tiles that chunk completes are partitioned while the next chunk crosses PCIe -- and
// only k_bucket_leaves is left behind the last chunk.  Decided before the first chunk (build_begin knows everything but
// the points); anything that is not a steady-state two-launch build goes the old way: transfer, then the whole build.  Same
// kernels over the same tiles: the grid is bit-identical either way (tests/test_gpu_handoff.py).
// MEASURED AND NOT THE DEFAULT (profiles/r05_handoff_chunk_pass_ab.txt, C3, alternating blocks of scans in one process): on
// the pull kernels' stream every partition launch stands between two transfers (a 32-tile launch is a 10-14 us chain with
// PCIe idle: 0.84 -> 0.92 ms per scan); on a stream of its own the two event hops around the last chunk's launch (pull ->
// partition -> leaves, ~20 us each on this runtime) cost more than the 14 us of partition they hide (0.80 -> 0.84 ms).
int build_grid_handoff(ndt_handle* h, const float* xyz, const float* x, const float* y, const float* z, size_t n, size_t stride) {
  h->build_pending = false;
  h->deferred_rc = 0;
  h->ms_settle_wait = 0;
  HIP_TRY(h, h->tx.ensure(n));
  HIP_TRY(h, h->ty.ensure(n));
  HIP_TRY(h, h->tz.ensure(n));
  ndt_handle::BuildRun& br = h->brun;
  int rc = build_begin(h, h->tx.p, h->ty.p, h->tz.p, n, br);
  if (rc) return rc;
  const bool steady = br.clean_cap != 0 && br.clean_cap == h->cell2leaf.cap;   // (build_enqueue's `optimistic`)
  br.pass_chunked = br.bucketed_ok && steady && tuning().handoff_chunk_pass != 0;
  br.pass_tiles = 0;
  hipError_t hook_err = hipSuccess;
  const size_t tile = br.pass_chunked ? bucket_tile_points(n) : 1;
  const int ntiles = br.pass_chunked ? bucket_build_tiles(n) : 0;
  // The partition launches go to a stream of their own, each behind its chunk's pull kernel by an event: on the pull
  // kernels' stream they would stand between two transfers (measured: PCIe idle for every partition launch, the scan
  // 0.84 -> 0.92 ms); the leaves launch joins them below.
  size_t ev_used = 0;
  if (br.pass_chunked && !h->pstream) HIP_TRY(h, hipStreamCreateWithFlags(&h->pstream, hipStreamNonBlocking));
  const std::function<void(size_t, bool)> hook = [&](size_t have, bool last) {
    const int upto = last ? ntiles : (int)(have / tile);
    if (upto <= br.pass_tiles || hook_err != hipSuccess) return;
    if (ev_used == h->pass_ev.size()) {
      hipEvent_t e = nullptr;
      if ((hook_err = hipEventCreateWithFlags(&e, hipEventDisableTiming)) != hipSuccess) return;
      h->pass_ev.push_back(e);
    }
    hipEvent_t e = h->pass_ev[ev_used++];
    if ((hook_err = hipEventRecord(e, h->stream)) != hipSuccess) return;
    if ((hook_err = hipStreamWaitEvent(h->pstream, e, 0)) != hipSuccess) return;
    hook_err = launch_bucket_pass_tiles(h->tx.p, h->ty.p, h->tz.p, n, br.inv_leaf, h->bucket_tab.p, h->stats.p, br.dirty_slots,
                                        h->cell2leaf.p, h->cell2leaf.cap, h->bnd.p, h->nleaf.p, h->xyz4.p, br.pass_tiles, upto,
                                        h->pstream);
    br.pass_tiles = upto;
    ++h->n_chunked_pass_launches;
  };
  rc = upload_soa(h, h->lane_t, h->stream, xyz, x, y, z, n, stride, h->tx, h->ty, h->tz, false, br.pass_chunked ? &hook : nullptr);
  if (rc == NDT_OK && hook_err == hipSuccess && br.pass_tiles > 0) {   // the join: `stream` goes on behind the last partition launch
    if (ev_used == h->pass_ev.size()) {
      hipEvent_t e = nullptr;
      if ((hook_err = hipEventCreateWithFlags(&e, hipEventDisableTiming)) == hipSuccess) h->pass_ev.push_back(e);
    }
    if (hook_err == hipSuccess) hook_err = hipEventRecord(h->pass_ev[ev_used], h->pstream);
    if (hook_err == hipSuccess) hook_err = hipStreamWaitEvent(h->stream, h->pass_ev[ev_used], 0);
  }
  if (rc == NDT_OK && hook_err != hipSuccess) rc = fail(h, NDT_ERR_HIP, hipGetErrorString(hook_err));
  if (rc) {
    if (br.pass_tiles > 0) {   // (the tiles enqueued so far have folded their bounds into the words)
      (void)hipStreamSynchronize(h->pstream);
      (void)neutral_bounds(h);
    }
    return rc;
  }
  if (br.pass_chunked) ++h->n_chunked_pass_builds;
  if (!br.pass_chunked) {   // the build's clock starts behind the transfer, as it does for build_grid
    br.t0 = std::chrono::steady_clock::now();
    if (br.build_events) HIP_TRY(h, hipEventRecord(h->ev0, h->stream));
  }
  rc = build_enqueue(h, br);
  if (rc) return rc;
  if (br.optimistic) {
    h->build_pending = true;
    return NDT_OK;
  }
  return build_complete(h, br);
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


}  // namespace engine
}  // namespace ndt

extern "C" {

int ndt_set_target(ndt_handle* h, const float* xyz, size_t n, size_t stride_bytes) {
  if (!h || (!xyz && n) || stride_bytes < 12 || stride_bytes % 4) return NDT_ERR_INVALID_ARG;
  int rc = bind_device(h);
  if (rc) return rc;
  settle_discard(h);
  const bool async = h->handoff_mode == NDT_HANDOFF_ASYNC;
  if (async && n > 0) return build_grid_handoff(h, xyz, nullptr, nullptr, nullptr, n, stride_bytes);
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
  if (async && n > 0) return build_grid_handoff(h, nullptr, x, y, z, n, 0);
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


}  // extern "C"
