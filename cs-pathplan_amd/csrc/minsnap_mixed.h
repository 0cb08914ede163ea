// minsnap_mixed.h -- device-resident bookkeeping of the mixed-order entry (minsnap_mixed.hip).
#pragma once
#include <stdint.h>

namespace csp {

constexpr int CSP_TRAJ_SKIPPED_BIT = 4;   // = CSP_TRAJ_SKIPPED of include/csp_minsnap.h

// length class k <-> 64 - k lanes per trajectory (= chunks of <= 4 segments: ceil(S / 4)), longest first
struct MixedTable {
    int32_t count[4 * 64];           // trajectories per (order - 2, length class)
    int32_t bucket_start[4][65];     // where bucket (order, class) starts in `perm`; [.][64] = the order's end
    int32_t unit_start[4][65];       // cumulative 64-lane work units of the order's classes; [.][64] = the order's total
    int32_t served;                  // trajectories that were bucketed (the rest carry CSP_TRAJ_SKIPPED)
};

}  // namespace csp
