// ndt_kernels.h -- launch wrappers of the HIP kernels (internal).
#pragma once

#include "ndt_device.h"

namespace ndt {

// ---- target voxel-grid build (ndt_target.hip) ------------------------------
struct FinalizeParams {
  double eig_ratio;
  int cov_mode;  // 0 svn, 1 pcl (recalled)
};
// One launch: per-block bounds rows (device memory, bounds_rows(n) x 8 ints), the fold of the
// rows + the grid geometry + the sort plan by the block that draws the last ticket (into *gd,
// device memory, and *gd_host, a pinned copy for the host to read at the end of the build),
// the reset of the cells the previous build published (old_stats / dirty_slots), and the
// zeroing of d_nleaf[0..1].  cell_capacity: cells the dense grid can hold (BG_CAPACITY beyond);
// planned_passes: digit passes the host has enqueued (0 = the device decides, host reads it).
// *ticket: zero-initialised, left at zero.
int bounds_rows(size_t n);
int sort_passes_for_cells(long long ncells);
void launch_bounds_geometry(const float* x, const float* y, const float* z, size_t n, float leaf, float inv_leaf,
                            long long cell_capacity, int planned_passes, int* rows, unsigned int* ticket,
                            BuildGeom* gd, BuildGeom* gd_host, const LeafStats* old_stats, int dirty_slots,
                            int* cell2leaf, size_t c2l_cap, int* d_nleaf, hipStream_t s);

// Cell key per point + xyz4 (n x 4 floats, 16-byte aligned: packed copy of the cloud for the
// per-voxel gather) + the first sort digit's tile histograms into sort_temp.  Geometry and
// sort plan are read from *gd (device memory); nothing runs if gd->status != BG_OK.
void launch_cell_keys(const float* x, const float* y, const float* z, size_t n, const BuildGeom* gd,
                      uint32_t* keys, float* xyz4, void* sort_temp, hipStream_t s);

// Stable LSD radix sort of (key, point index) on the key bits of the plan in *gd; values start
// as the identity.  Ping-pongs between the a and b buffers; *result_in_b tells where the result is.
size_t sort_temp_bytes(size_t n);
hipError_t sort_pairs(void* temp, uint32_t* keys_a, uint32_t* keys_b, uint32_t* vals_a, uint32_t* vals_b,
                      size_t n, int passes, const BuildGeom* gd, hipStream_t s, bool* result_in_b);
// test seam: a plan for keys of end_bit bits (host struct), and the first digit's histograms
// for keys that did not come from launch_cell_keys
void fill_sort_plan(BuildGeom* b, int end_bit);
void launch_sort_first_count(const uint32_t* keys, size_t n, const BuildGeom* gd, void* sort_temp, hipStream_t s);

// One launch per digit, straight from the cloud (no launch_cell_keys): see k_sort_pass.  Only for
// clouds with fused_sort_fits(n, CUs of the device); `table`: fused_table_words() words, zero at allocation; a build
// that had to give up waiting leaves BG_SPIN in *gd / *gd_host (repeat it with the classic passes).
bool fused_build_enabled();  // NDT_FUSED_SORT != 0 (default): fused sort passes and the fused run search
bool fused_sort_fits(size_t n, int compute_units);
int fused_tile_for(size_t n, int compute_units);  // pairs per tile (8192 up to 2 M points on 256 CUs, 16384 up to 4 M), 0 = classic passes
size_t fused_table_words();
hipError_t sort_cloud_fused(const float* x, const float* y, const float* z, size_t n, int tile, BuildGeom* gd,
                            BuildGeom* gd_host, float* xyz4, uint32_t* keys_a, uint32_t* keys_b, uint32_t* vals_a,
                            uint32_t* vals_b, int passes, uint32_t* table, uint32_t* seq, hipStream_t s,
                            bool* result_in_b);

// The whole build in TWO launches (k_bucket_pass + k_bucket_leaves, ndt_target.hip): steady state only --
// every buffer exists and the dense grid holds nothing but the `dirty_slots` cells of old_stats.  tab: the
// partition's column table, bucket_table_words() words (written by the first launch, read by the second: no
// tags, no initial state); pts4: n float4 (the cloud, tile by tile, each tile in bucket order); sums: 9 doubles
// per leaf slot.  Neither launch waits for sibling blocks.  Refusals leave BG_BUCKET / BG_CAPACITY / ... in
// *gd_host: decided before anything is written, except a bucket beyond a block's LDS or hash table (late BG_BUCKET).
bool bucket_build_enabled();                            // NDT_BUCKET_BUILD != 0 (default on)
// the 8 bounds words {min xyz, max xyz, #finite, unused} between two builds (host copy for (re)initialisation;
// the launch pair leaves them neutral again whenever it runs to its end)
void bucket_bounds_neutral(int out[8]);
bool bucket_build_fits(size_t n, int compute_units);
int bucket_build_tiles(size_t n);
size_t bucket_table_words();
// ... or piecewise (the asynchronous host hand-off runs the partition under the transfer, chunk by chunk, and only the
// leaves launch behind the last chunk): tiles [tile_first, tile_end) of bucket_build_tiles(n), each of
// bucket_tile_points(n) points; every tile exactly once, tile 0's launch first (it resets the previous build's cells)
size_t bucket_tile_points(size_t n);
hipError_t launch_bucket_pass_tiles(const float* x, const float* y, const float* z, size_t n, float inv_leaf, uint32_t* tab,
                                    const LeafStats* old_stats, int dirty_slots, int* cell2leaf, size_t c2l_cap, int* bnd,
                                    int* d_nleaf, float* pts4, int tile_first, int tile_end, hipStream_t s);
hipError_t launch_bucket_leaves(size_t n, float leaf, float inv_leaf, long long cell_capacity, int min_pts, FinalizeParams fp,
                                BuildGeom* gd, BuildGeom* gd_host, const uint32_t* tab, int* cell2leaf, int* bnd, int* d_nleaf,
                                unsigned int* ticket, const float* pts4, double* sums, VoxelRecord* rec, float* cent4,
                                LeafStats* stats, int max_leaves, int* nleaf_host, int done_tag, hipStream_t s);
hipError_t launch_bucket_build(const float* x, const float* y, const float* z, size_t n, float leaf, float inv_leaf,
                               long long cell_capacity, int min_pts, FinalizeParams fp, BuildGeom* gd, BuildGeom* gd_host,
                               uint32_t* tab, const LeafStats* old_stats, int dirty_slots, int* cell2leaf,
                               size_t c2l_cap, int* bnd /* 8 ints, see bucket_bounds_neutral */, int* d_nleaf,
                               unsigned int* ticket, float* pts4,
                               double* sums, VoxelRecord* rec, float* cent4 /* 4 floats per leaf slot: f32 centroid + chain link */,
                               LeafStats* stats, int max_leaves, int* nleaf_host, int done_tag, hipStream_t s);

// runs of equal cell key with >= min_pts points get a leaf slot (ascending cell order);
// block_counts / block_offsets: runs_blocks(n) ints each; d_nleaf[0] receives the total
// (count pass; its last block also scans the counts); *ticket as above
int runs_blocks(size_t n);
// run_tags != nullptr (run_tag_words(n) words, zero at allocation; *seq its tag counter): count and
// emit in ONE launch; a block that had to give up waiting leaves BG_SPIN in *gd / *gd_host
size_t run_tag_words(size_t n);
hipError_t launch_find_runs(const uint32_t* keys_sorted, size_t n, BuildGeom* gd, BuildGeom* gd_host, int min_pts,
                            int* d_nleaf, int* block_counts, int* block_offsets, unsigned int* ticket, uint32_t* run_tags,
                            size_t run_tags_cap, uint32_t* seq, int* leaf_start, int* leaf_cnt, hipStream_t s);

// tmp: the cloud chunk by chunk as [x | y | z] of `chunk` points (the last chunk shorter) -> SoA arrays
// host hand-off: one chunk [x(seg) | y(seg) | z(seg)] in mapped pinned host memory -> the SoA arrays (pulled over PCIe)
void launch_pull_chunk(const float* stage_dev, size_t len, size_t seg, float* x, float* y, float* z, hipStream_t s);
// pcl::VoxelGrid centroids of the runs found by launch_find_runs(min_pts = 1): output point r = float mean of run r's
// points (and of `intensity`, may be null; oi may be null), ascending voxel index; at most `cap` points are written
void launch_voxel_centroids(const float* xyz4, const float* intensity, const uint32_t* vals_sorted, const int* d_nleaf,
                            const int* leaf_start, const int* leaf_cnt, size_t max_runs, size_t cap, float* ox, float* oy,
                            float* oz, float* oi, hipStream_t s);
// multi-grid union table: cell2leaf[cells[i]] = slots[i], i < n (device arrays)
void launch_scatter_heads(const int* cells, const int* slots, size_t n, int* cell2leaf, hipStream_t s);
int finalize_blocks(int max_leaves);
int build_read_stamps(unsigned long long* out /* 4 x 512 x 8 */);  // -DNDT_STAMPS builds only; else 0
// per-leaf sums, then per-leaf statistics; sums: 9 doubles per leaf slot (scratch)
void launch_finalize_leaves(const float* xyz4, const uint32_t* keys_sorted, const uint32_t* vals_sorted,
                            int* d_nleaf /* [0]=slots, [1]=valid */, const int* leaf_start, const int* leaf_cnt,
                            int max_leaves, FinalizeParams fp, double* sums, VoxelRecord* rec,
                            float* cent4 /* 4 floats per leaf slot: the f32 centroid the radius search tests + chain link */,
                            LeafStats* stats, int* cell2leaf, int* block_ok /* finalize_blocks(max_leaves) ints, scratch */,
                            unsigned int* ticket /* zero, left at zero */,
                            int* nleaf_host /* pinned, 16-byte aligned: receives {d_nleaf[0], d_nleaf[1], done_tag, 0} in one store */,
                            int done_tag, hipStream_t s);

// out[i] = (float)(R x + t) in f64 (sliding-window target assembly); out arrays hold n floats
// device-to-device copy of three SoA arrays in one launch
void launch_copy_soa(const float* x, const float* y, const float* z, size_t n, float* ox, float* oy, float* oz,
                     hipStream_t s);
void launch_transform_append(const float* x, const float* y, const float* z, size_t n,
                             const double pose_colmajor[16], float* ox, float* oy, float* oz,
                             hipStream_t s);

// Source ordering: the source cloud stably sorted by the 8 x 8 x 4-voxel block of the target grid
// its points fall into under the transform P (keys + first histogram, the digit passes, gather).
// keys/vals a, b: n uint32 each; temp: sort_temp_bytes(n); plan: one BuildGeom of scratch in
// device memory; ox/oy/oz: the sorted copy.
int source_sort_passes(const GridGeom& g);
hipError_t sort_source_by_blocks(const float* x, const float* y, const float* z, size_t n, const GridGeom& g,
                                 const PoseConsts& P, BuildGeom* plan, void* temp, uint32_t* keys_a,
                                 uint32_t* keys_b, uint32_t* vals_a, uint32_t* vals_b, float* ox, float* oy,
                                 float* oz, hipStream_t s);

// ---- derivative evaluation (ndt_derivs.hip) ---------------------------------
// cus: compute units of the handle's device (EvalConsts::compute_units): single-pose launches of mid-sized scans are
// shaped one block per CU, and the XCD count follows from it
int derivs_grid_blocks(size_t n_src, int K, int cus);
int derivs_block_threads(size_t n_src, int K, int cus);
size_t derivs_partials_words(size_t n_src, int K, int cus);  // doubles needed in d_partials
int derivs_counters_per_pose();
// which finishing wave expands which wave's points: 2 bits per wave (owning SIMD), 4 bits per SIMD (its finishing wave)
void derivs_item_owners(int threads, unsigned int* owners_out, unsigned int* fin_waves_out, int* lone_out = nullptr);
int derivs_read_wave_stamps(unsigned long long* out, int nblocks);  // diagnostic builds only: per-wave stamps, see ndt_derivs.hip
int derivs_read_stamps(unsigned long long* out, int nblocks);  // diagnostic builds (-DNDT_STAMPS) only                     // ticket words per pose in d_counters
// d_partials: derivs_partials_words() doubles, ZEROED when allocated (rows of tagged slots);
// d_counters: K * derivs_counters_per_pose() zero-initialised ticket words (left at zero again
// by every launch).  `seq`: a sequence number no earlier launch IN THIS PROCESS has used (it
// tags every partial slot).  If d_poses is null the single pose `pose` is passed as a kernel
// argument.  One launch: the last block to finish adds the per-block rows in fixed order.
// Result: d_host_slots == nullptr -> K * EV_WORDS plain doubles in d_out (device memory);
// d_host_slots != nullptr (device-mapped pinned host memory, K * 2 * EV_WORDS words) -> 32 slots
// {seq, value} per pose for the host to poll, d_out unused.
// d_mbox != nullptr (single-pose only): a PRE-LAUNCHED evaluation -- `pose` is ignored, the kernel
// waits for the pose to appear in *d_mbox under its sequence number (PoseMailbox in ndt_device.h).
void launch_derivatives(const float* sx, const float* sy, const float* sz, size_t n_src,
                        const GridGeom& g, const int* cell2leaf /* readable 4 ints beyond either end */,
                        const VoxelRecord* rec, const float* cent4 /* per-leaf f32 centroid + chain link (KDTREE, multi-grid) */,
                        const PoseConsts& pose, const PoseConsts* d_poses, int K,
                        const EvalConsts& ec, double* d_partials, unsigned int* d_counters,
                        double* d_out, hipStream_t s, unsigned long long* d_host_slots,
                        unsigned long long seq, const PoseMailbox* d_mbox = nullptr,
                        // NDT_REDUCE_P2P (single-pose launches): the block that finishes the local sum
                        // exchanges it with the other ranks under the tag `xround` (XchgInfo, ndt_device.h)
                        const XchgInfo* d_xinfo = nullptr, unsigned long long xround = 0ull,
                        // pre-launched launches: a zeroed device counter and a pinned host word that receives
                        // `seq` once every block of the launch is resident
                        unsigned int* d_arrive_ctr = nullptr, unsigned long long* d_arrived_host = nullptr,
                        // kernel timing: events attached to the dispatch itself (begin / end of the kernel, as rocprofv3
                        // reports it); both or neither
                        hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr,
                        // single-pose ordinary launches enqueued behind the build of their own grid: the geometry is read
                        // from the build's device-side BuildGeom (and nothing runs after a refused build), `g` is ignored
                        const BuildGeom* d_geom = nullptr);

// The 80-byte records of leaf slots [0, n) as 48-byte PackedRecords (f64 mean, f32 inverse covariance); a launch
// whose EvalConsts::packed is set takes that array in place of `rec`.
void launch_pack_records(const VoxelRecord* rec, PackedRecord* out, size_t n, hipStream_t s);

void launch_transform(const float* sx, const float* sy, const float* sz, size_t n,
                      const PoseConsts& pose, float* out_xyz, hipStream_t s);

}  // namespace ndt
