// Compatibility header (ref: include/registercallback.hpp:7-10): everything pclomp's four NDT
// headers declare that the reference's drivers use lives in ndt_omp.h of this directory.
#pragma once
#include "ndt_omp.h"
