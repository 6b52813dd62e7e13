// ndt_tuning.cpp -- see ndt_tuning.h.
#include "ndt_tuning.h"

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>

namespace ndt {

void tuning_defaults(ndt_tuning* t) {
  std::memset(t, 0, sizeof(*t));
  t->deriv_block = 0;
  t->deriv_summer = 1;
  t->deriv_dedicated = 1;
  t->deriv_single_level_max = 2048;
  t->deriv_xcd = 1;
  t->bucket_build = 1;
  t->bucket_tile = 0;
  t->fused_sort = 1;
  t->bounds_blocks = 256;
  t->bounds_unroll = 8;
  t->finalize_threads = 256;
  t->build_events = -1;
  t->build_wait_sync = 0;
  t->mbox_tagged = 1;
  t->mbox_preload = 0;
  t->prelaunch_streams = 2;
  t->prelaunch_probe = 1;
  t->speculate_first = 1;
  t->timing_bracket = 0;
  t->handoff_chunk_pass = 0;
  t->deriv_summer_split = 1;
  t->deriv_one_block_per_cu = 1;
}

namespace {

bool valid(const ndt_tuning& t) {
  auto flag = [](int v) { return v == 0 || v == 1; };
  if (t.deriv_block != 0 && (t.deriv_block < 64 || t.deriv_block > 1024 || t.deriv_block % 64 != 0)) return false;
  if (!flag(t.deriv_summer) || !flag(t.deriv_dedicated)) return false;
  if (t.deriv_single_level_max < 1) return false;
  if (t.deriv_xcd < 0 || t.deriv_xcd > 2) return false;
  if (!flag(t.bucket_build) || !flag(t.fused_sort)) return false;
  if (t.bucket_tile != 0 && t.bucket_tile != 1024 && t.bucket_tile != 2048 && t.bucket_tile != 4096 && t.bucket_tile != 8192) return false;
  if (t.bounds_blocks < 1) return false;
  if (t.bounds_unroll != 4 && t.bounds_unroll != 8) return false;
  if (t.finalize_threads != 64 && t.finalize_threads != 256) return false;
  if (t.build_events < -1 || t.build_events > 1) return false;
  if (!flag(t.build_wait_sync) || !flag(t.mbox_tagged) || !flag(t.mbox_preload)) return false;
  if (t.prelaunch_streams != 1 && t.prelaunch_streams != 2) return false;
  if (!flag(t.prelaunch_probe) || !flag(t.speculate_first) || !flag(t.timing_bracket) || !flag(t.handoff_chunk_pass) || !flag(t.deriv_one_block_per_cu)) return false;
  if (t.deriv_summer_split != 0 && t.deriv_summer_split != 1 && t.deriv_summer_split != 4 && t.deriv_summer_split != 8) return false;
  for (int v : t.reserved)
    if (v != 0) return false;
  return true;
}

// Written rarely (a tuning program between two runs), read at every launch: the struct is copied under a mutex on both
// sides -- 128 bytes, no launch path takes it more than once.
std::mutex g_mu;
ndt_tuning g_tuning;
std::once_flag g_once;

void init_once() {
  std::call_once(g_once, [] {
    tuning_defaults(&g_tuning);
#ifdef NDT_TUNING_ENV
    // diagnostic library variants only: the historical variable names as INITIAL values
    auto env = [](const char* name, int* field) {
      const char* e = std::getenv(name);
      if (e && *e) *field = std::atoi(e);
    };
    ndt_tuning t = g_tuning;
    env("NDT_DERIV_BLOCK", &t.deriv_block);
    env("NDT_DERIV_SUMMER", &t.deriv_summer);
    env("NDT_DERIV_DEDICATED", &t.deriv_dedicated);
    env("NDT_DERIV_SINGLE_LEVEL_MAX", &t.deriv_single_level_max);
    env("NDT_DERIV_XCD", &t.deriv_xcd);
    env("NDT_BUCKET_BUILD", &t.bucket_build);
    env("NDT_BUCKET_TILE", &t.bucket_tile);
    env("NDT_FUSED_SORT", &t.fused_sort);
    env("NDT_BOUNDS_BLOCKS", &t.bounds_blocks);
    env("NDT_BOUNDS_UNROLL", &t.bounds_unroll);
    env("NDT_FINALIZE_THREADS", &t.finalize_threads);
    env("NDT_BUILD_EVENTS", &t.build_events);
    if (const char* e = std::getenv("NDT_BUILD_WAIT")) t.build_wait_sync = std::strcmp(e, "sync") == 0 ? 1 : 0;
    env("NDT_MBOX_TAGGED", &t.mbox_tagged);
    env("NDT_MBOX_PRELOAD", &t.mbox_preload);
    env("NDT_PRELAUNCH_STREAMS", &t.prelaunch_streams);
    env("NDT_PRELAUNCH_PROBE", &t.prelaunch_probe);
    env("NDT_SPECULATE_FIRST", &t.speculate_first);
    env("NDT_TIMING_BRACKET", &t.timing_bracket);
    env("NDT_HANDOFF_CHUNK_PASS", &t.handoff_chunk_pass);
    env("NDT_DERIV_SUMMER_SPLIT", &t.deriv_summer_split);
    env("NDT_DERIV_ONE_BLOCK_PER_CU", &t.deriv_one_block_per_cu);
    if (valid(t)) g_tuning = t;
#endif
  });
}

}  // namespace

const ndt_tuning& tuning() {
  // (a thread-local snapshot: callers keep the reference for the length of a call)
  static thread_local ndt_tuning snap;
  init_once();
  std::lock_guard<std::mutex> lk(g_mu);
  snap = g_tuning;
  return snap;
}

int tuning_set(const ndt_tuning* t) {
  if (!t || !valid(*t)) return NDT_ERR_INVALID_ARG;
  init_once();
  std::lock_guard<std::mutex> lk(g_mu);
  g_tuning = *t;
  return NDT_OK;
}

}  // namespace ndt

extern "C" {

int ndt_get_tuning(ndt_tuning* out) {
  if (!out) return NDT_ERR_INVALID_ARG;
  *out = ndt::tuning();
  return NDT_OK;
}

int ndt_set_tuning(const ndt_tuning* t) { return ndt::tuning_set(t); }

}  // extern "C"
