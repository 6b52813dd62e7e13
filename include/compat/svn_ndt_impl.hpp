// Compatibility header (ref: include/registercallback.hpp:14-17): the svn_ndt engine the
// drivers use is declared in svn_ndt.h of this directory; there is no separate implementation
// or voxel-grid header to include (the grid lives on the device).
#pragma once
#include "svn_ndt.h"
