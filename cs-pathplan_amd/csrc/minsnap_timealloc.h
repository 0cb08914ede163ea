// minsnap_timealloc.h -- time allocation fused with the re-solve loop's initial state (minsnap_timealloc.hip).
#pragma once
#include "minsnap_launch.h"

namespace csp {

// csp_minsnap_plan_batch / csp_minsnap_generate_batch with path_weight > 0: the segment times (minimum_snap.cpp:59-72)
// and vw[b] = vw0, iters[b] = 0, done[b] = 0, *pending = 0 (the state launch_resolve_init sets) in ONE launch -- for one
// flight every launch of a one-wave kernel costs ~5 us whatever it does.
hipError_t launch_time_alloc_init(const TimeAllocArgs &a, bool f32, double *vw, int32_t *iters, int32_t *done, int32_t *pending,
                                  double vw0, hipStream_t st);

}  // namespace csp
