// sm_kernels.h -- hand-written gfx950 kernels of the per-frame fusion hot path.
// Included once by sm_api.hip.  Shader citations: /root/reference/src/Shaders/<file>:<line>.
//
// Frame images are held COLUMN-MAJOR (q = i*H + j, "x-outer / y-inner"): that is the order
// in which the reference submits pixels to data.vert (src/GlobalModel.cpp:67-74) and hence the
// order in which new surfels are appended, so association + ordered compaction become a 1-D
// coalesced stream over q.
#pragma once

#include "sm_device.h"

namespace sm {

#include "sm_k_prep.h"
#include "sm_k_cull.h"
#include "sm_k_pass.h"
#include "sm_k_assoc.h"
#include "sm_k_shard.h"
#include "sm_k_aux.h"

}  // namespace sm
