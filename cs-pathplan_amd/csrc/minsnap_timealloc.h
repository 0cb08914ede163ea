// minsnap_timealloc.h -- launches that fold the re-solve loop's bookkeeping into a neighbouring kernel: time allocation +
// the loop's initial state (minsnap_timealloc.hip), sample placement + the first pass's update (minsnap_plan.hip).
#pragma once
#include "minsnap_launch.h"

namespace csp {

// csp_minsnap_plan_batch / csp_minsnap_generate_batch with path_weight > 0: the segment times (minimum_snap.cpp:59-72)
// and vw[b] = vw0, iters[b] = 0, done[b] = 0, *pending = 0 (the state launch_resolve_init sets) in ONE launch -- for one
// flight every launch of a one-wave kernel costs ~5 us whatever it does.
hipError_t launch_time_alloc_init(const TimeAllocArgs &a, bool f32, double *vw, int32_t *iters, int32_t *done, int32_t *pending,
                                  double vw0, hipStream_t st);

// The per-segment wave sampler (launch_sample_segment_waves) with the bookkeeping of the re-solve loop's FIRST pass
// (resolve_update_kernel: raise the weight or mark the trajectory done, minimum_snap.cpp:80-90) done by its placement
// kernel -- one workgroup per trajectory anyway -- instead of a launch of its own.  `upd` may be null.
struct LoopUpdate {
    const double *max_dev;
    double *vw;
    int32_t *iters, *done, *pending;
};
hipError_t launch_sample_segment_waves_upd(const SampleArgs &a, double *tmp, const int64_t *tmp_off, int32_t *seg_counts,
                                           int64_t total_segments, const LoopUpdate *upd, hipStream_t st);

}  // namespace csp
