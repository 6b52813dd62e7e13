// Compatibility header: `#include <pclomp/ndt_omp.h>` resolving to the MI355X engine.
//
// The reference includes the un-vendored tier4/ndt_omp headers by these names
// (ref: include/registercallback.hpp:7-10) and names `pclomp::NormalDistributionsTransform`,
// `pclomp::DIRECT1 / DIRECT7 / KDTREE` and `pclomp::VoxelGridCovariance<PointT>` in its drivers
// (ref: run/pipeline.cpp:464-480, run/pipeline_ligo_tc.cpp:287-304,
//  run/pipeline_ins_map_distribution.cpp:346-365, include/pipeline.hpp:175).  With
// `include/compat` placed BEFORE `extern/ndt_omp/include` on the include path those names are
// the HIP engine's and the drivers compile unchanged.  Written from the call sites; nothing
// here comes from ndt_omp (its sources are absent from the reference tree).
#pragma once

#include "../../ndt_hip/ndt_hip.hpp"

namespace pclomp {

using NeighborSearchMethod = ndt_hip::NeighborSearchMethod;
constexpr NeighborSearchMethod KDTREE = ndt_hip::KDTREE;
constexpr NeighborSearchMethod DIRECT26 = ndt_hip::DIRECT26;
constexpr NeighborSearchMethod DIRECT7 = ndt_hip::DIRECT7;
constexpr NeighborSearchMethod DIRECT1 = ndt_hip::DIRECT1;

using NdtResult = ndt_hip::NdtResult;

template <typename PointSource, typename PointTarget>
using NormalDistributionsTransform = ndt_hip::NormalDistributionsTransform<PointSource, PointTarget>;

// getTargetCells() hands out one grid type whatever the point type
template <typename PointT>
using VoxelGridCovariance = ndt_hip::TargetGrid;

}  // namespace pclomp
