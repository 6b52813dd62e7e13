// ndt_tuning.h -- the engine's tuning / A-B switches (internal).  One process-wide ndt_tuning (include/ndt_hip.h);
// the production library changes it only through ndt_set_tuning(), the diagnostic variants (-DNDT_TUNING_ENV:
// make VARIANT=ab|seams|stamps) also seed it from the historical NDT_* environment variables at first use.
#pragma once

#include "../../include/ndt_hip.h"

namespace ndt {

const ndt_tuning& tuning();
void tuning_defaults(ndt_tuning* t);
int tuning_set(const ndt_tuning* t);  // NDT_OK | NDT_ERR_INVALID_ARG

}  // namespace ndt
