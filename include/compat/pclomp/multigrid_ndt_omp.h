// Compatibility header: `#include <pclomp/multigrid_ndt_omp.h>` resolving to the MI355X engine.
//
// The reference's build names tier4/ndt_omp's multi-grid sources (ref: CMakeLists.txt:41-42) but
// its tree holds neither them nor a caller; the class below is [RECALLED] from tier4's public
// interface (addTarget / removeTarget / createVoxelKdtree with string ids) and is the same engine
// object as pclomp::NormalDistributionsTransform.  Nothing here comes from ndt_omp.
#pragma once

#include "ndt_omp.h"

namespace pclomp {

template <typename PointSource, typename PointTarget>
using MultiGridNormalDistributionsTransform = ndt_hip::MultiGridNormalDistributionsTransform<PointSource, PointTarget>;

}  // namespace pclomp
