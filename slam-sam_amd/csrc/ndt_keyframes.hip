// ndt_keyframes.hip -- multi-grid targets, device-resident keyframe archive, voxel downsample (see ndt_engine.h).
#include "ndt_engine.h"

extern "C" {

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


}  // extern "C"
