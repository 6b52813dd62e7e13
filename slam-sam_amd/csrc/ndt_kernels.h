// ndt_kernels.h -- launch wrappers of the HIP kernels (internal).
#pragma once

#include "ndt_device.h"

namespace ndt {

// ---- target voxel-grid build (ndt_target.hip) ------------------------------
// rows: bounds_rows(n) x 8 ints = ordered-int encoded {min x,y,z, max x,y,z}, #finite, 0
// per block (device-visible memory); fold_bounds() merges them on the host.
int bounds_rows(size_t n);
void launch_bounds(const float* x, const float* y, const float* z, size_t n, int* rows, hipStream_t s);
void fold_bounds(const int* rows, int nrows, int out[8]);
float decode_ordered(int enc);

// Cell key per point + xyz4 (n x 4 floats, 16-byte aligned: packed copy of the cloud for the
// per-voxel gather) + the first sort digit's tile histograms into sort_temp.
void launch_cell_keys(const float* x, const float* y, const float* z, size_t n, const GridGeom& g,
                      uint32_t* keys, float* xyz4, int end_bit, void* sort_temp, hipStream_t s);

// Stable LSD radix sort of (key, point index) on the low end_bit bits; values start as the
// identity.  Ping-pongs between the a and b buffers; *result_in_b tells where the result is.
size_t sort_temp_bytes(size_t n);
hipError_t sort_pairs(void* temp, uint32_t* keys_a, uint32_t* keys_b, uint32_t* vals_a, uint32_t* vals_b,
                      size_t n, int end_bit, hipStream_t s, bool* result_in_b);
// test seam: the first digit's histograms for keys that did not come from launch_cell_keys
void launch_sort_first_count(const uint32_t* keys, size_t n, int end_bit, void* sort_temp, hipStream_t s);

// runs of equal cell key with >= min_pts points get a leaf slot (ascending cell order);
// block_counts / block_offsets: runs_blocks(n) ints each; d_nleaf[0] receives the total
int runs_blocks(size_t n);
void launch_find_runs(const uint32_t* keys_sorted, size_t n, int ncells, int min_pts, int* d_nleaf,
                      int* block_counts, int* block_offsets, int* leaf_start, int* leaf_cnt,
                      hipStream_t s);

// resets cell2leaf[stats[slot].cell] = -1 for slot < n_slots (the cells the last build published)
void launch_clear_cells(const LeafStats* stats, int n_slots, int* cell2leaf, size_t cap, hipStream_t s);

struct FinalizeParams {
  double eig_ratio;
  int cov_mode;  // 0 svn, 1 pcl (recalled)
};
// sums: 9 doubles per leaf slot (scratch)
void launch_finalize_leaves(const float* xyz4,
                            const uint32_t* keys_sorted, const uint32_t* vals_sorted,
                            int* d_nleaf /* [0]=slots, [1]=valid */, const int* leaf_start,
                            const int* leaf_cnt, int max_leaves, FinalizeParams fp, double* sums,
                            VoxelRecord* rec, LeafStats* stats, int* cell2leaf, hipStream_t s);

// out[i] = (float)(R x + t) in f64 (sliding-window target assembly); out arrays hold n floats
// device-to-device copy of three SoA arrays in one launch
void launch_copy_soa(const float* x, const float* y, const float* z, size_t n, float* ox, float* oy, float* oz,
                     hipStream_t s);
void launch_transform_append(const float* x, const float* y, const float* z, size_t n,
                             const double pose_colmajor[16], float* ox, float* oy, float* oz,
                             hipStream_t s);

// ---- derivative evaluation (ndt_derivs.hip) ---------------------------------
int derivs_grid_blocks(size_t n_src, int K);
int derivs_block_threads(size_t n_src, int K);
size_t derivs_partials_words(size_t n_src, int K);  // doubles needed in d_partials
int derivs_counters_per_pose();
int derivs_read_stamps(unsigned long long* out, int nblocks);  // diagnostic builds (-DNDT_STAMPS) only                     // ticket words per pose in d_counters
// d_partials: derivs_partials_words() doubles, ZEROED when allocated (rows of tagged slots);
// d_counters: K * derivs_counters_per_pose() zero-initialised ticket words (left at zero again
// by every launch).  `seq`: a sequence number no earlier launch IN THIS PROCESS has used (it
// tags every partial slot).  If d_poses is null the single pose `pose` is passed as a kernel
// argument.  One launch: the last block to finish adds the per-block rows in fixed order.
// Result: d_host_slots == nullptr -> K * EV_WORDS plain doubles in d_out (device memory);
// d_host_slots != nullptr (single-pose path only, device-mapped pinned host memory, 2 * EV_WORDS
// words) -> 32 slots {seq, value} for the host to poll, d_out unused.
void launch_derivatives(const float* sx, const float* sy, const float* sz, size_t n_src,
                        const GridGeom& g, const int* cell2leaf, const VoxelRecord* rec,
                        const PoseConsts& pose, const PoseConsts* d_poses, int K,
                        const EvalConsts& ec, double* d_partials, unsigned int* d_counters,
                        double* d_out, hipStream_t s, unsigned long long* d_host_slots,
                        unsigned long long seq);

void launch_transform(const float* sx, const float* sy, const float* sz, size_t n,
                      const PoseConsts& pose, float* out_xyz, hipStream_t s);

}  // namespace ndt
