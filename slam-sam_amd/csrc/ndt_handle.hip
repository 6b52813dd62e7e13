// ndt_handle.hip -- handle life cycle, parameters, reducer entry points, timing, test seams (see ndt_engine.h).
#include "ndt_engine.h"

namespace ndt {
namespace engine {

int fail(ndt_handle* h, int code, const std::string& msg) {
  if (h) h->err = msg;
  return code;
}


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


}  // namespace engine
}  // namespace ndt

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
  if (h->pstream) (void)hipStreamSynchronize(h->pstream);
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
  h->run_counts.release(); h->run_offsets.release(); h->fin_counts.release(); h->bucket_tab.release(); h->bnd.release(); h->sort_tags.release(); h->run_tags.release(); h->xyz4.release(); h->leaf_sums.release();
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
  if (h->pstream) (void)hipStreamDestroy(h->pstream);
  for (hipEvent_t e : h->pass_ev) (void)hipEventDestroy(e);
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

// test seam (not in the public header): host hand-offs whose partition ran under the transfer {builds, launches of tile ranges}
int ndt_debug_handoff_counters(const ndt_handle* h, int64_t out[2]) {
  if (!h || !out) return NDT_ERR_INVALID_ARG;
  out[0] = h->n_chunked_pass_builds;
  out[1] = h->n_chunked_pass_launches;
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

// test seam (not in the public header): the shape of a k_derivatives launch over n_src points and K poses on a device of `cus`
// compute units -- {threads per block, blocks that own points, summing blocks in front of them, blocks of the grid (per pose)}
int ndt_debug_launch_shape(size_t n_src, int K, int cus, int out[4]) {
  if (!out || K < 1 || cus < 1) return NDT_ERR_INVALID_ARG;
  const int threads = derivs_block_threads(n_src, K, cus);
  const int grid = derivs_grid_blocks(n_src, K, cus);
  const int pb = (int)std::max<size_t>(1, (n_src + (size_t)threads - 1) / (size_t)threads);
  out[0] = threads;
  out[1] = pb;
  out[2] = grid - pb;
  out[3] = grid;
  return NDT_OK;
}

// test seam (not in the public header): the two-launch build's partition plan for a cloud of n points --
// {fits (0 / 1), points per tile, tiles, words of the column table}
int ndt_debug_bucket_plan(size_t n, long long out[4]) {
  if (!out) return NDT_ERR_INVALID_ARG;
  const bool fits = bucket_build_fits(n, 0);
  out[0] = fits ? 1 : 0;
  out[1] = fits ? (long long)bucket_tile_points(n) : 0;
  out[2] = fits ? bucket_build_tiles(n) : 0;
  out[3] = (long long)bucket_table_words();
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
