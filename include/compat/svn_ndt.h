// Compatibility header: `#include <svn_ndt.h>` resolving to the MI355X engine
// (ref: include/registercallback.hpp:14-17; driver run/pipeline_lo_svn.cpp:299-320,387-388;
//  the interface restated is extern/svn_ndt/include/svn_ndt.h:30-51,100-182).  Put
// `include/compat` BEFORE `extern/svn_ndt/include` on the include path.
#pragma once

#include "../ndt_hip/ndt_hip.hpp"

namespace svn_ndt {

using NeighborSearchMethod = ndt_hip::SvnNeighborSearchMethod;
using SvnNdtResult = ndt_hip::SvnNdtResult;

template <typename PointSource, typename PointTarget>
using SvnNormalDistributionsTransform = ndt_hip::SvnNormalDistributionsTransform<PointSource, PointTarget>;

// svn_ndt::VoxelGridCovariance<PointT> (ref: extern/svn_ndt/include/voxel_grid_covariance.h): the host-side view of the
// device's voxel grid -- getLeaf(index | point), getLeaves, getCentroids, nearestKSearch, radiusSearch,
// getNeighborhoodAtPoint7 / 1, getMinPointPerVoxel, getCovEigValueInflationRatio; one grid type whatever the point type
template <typename PointT>
using VoxelGridCovariance = ndt_hip::TargetGrid;

}  // namespace svn_ndt
