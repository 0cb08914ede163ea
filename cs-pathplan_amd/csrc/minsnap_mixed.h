// minsnap_mixed.h -- device-resident bookkeeping of the mixed-order entry (minsnap_mixed.hip).
#pragma once
#include <stdint.h>

namespace csp {

constexpr int CSP_TRAJ_SKIPPED_BIT = 4;   // = CSP_TRAJ_SKIPPED of include/csp_minsnap.h

constexpr int MIXED_NCLS = 64;    // length classes per group
constexpr int MIXED_NGRP = 8;     // group = family * 4 + (order - 2)
// family 0 (minsnap_twist_impl.h, one lane pair per trajectory): class k <-> S = 64 - k segments exactly, longest first;
//          a work unit = 64 trajectories of one class;
// family 1 (minsnap_chunked_impl.h, 65 .. 256 segments): class k <-> 64 - k lanes per trajectory (= chunks of <= 4 segments:
//          ceil(S / 4)), longest first; a work unit = 64 lanes.
struct MixedTable {
    int32_t count[MIXED_NGRP * MIXED_NCLS];            // trajectories per (group, length class)
    int32_t bucket_start[MIXED_NGRP][MIXED_NCLS + 1];  // where bucket (group, class) starts in `perm`; [.][64] = the group's end
    int32_t unit_start[MIXED_NGRP][MIXED_NCLS + 1];    // cumulative work units of the group's classes; [.][64] = the group's total
    int32_t served;                                    // trajectories that were bucketed (the rest carry CSP_TRAJ_SKIPPED)
    // family 0 runs as ONE launch over all orders: its (order, S) classes in descending cost (TwistCostOrder), units
    // cumulated in that order; next_unit: the work counter of that launch (zeroed by the planning kernel)
    int32_t tw_ustart[4 * MIXED_NCLS + 1];
    int32_t next_unit;
};

// The lane-pair sweep's classes by descending cost of one work unit (~ per-step time of the order x segments): the heaviest
// units start first and the rest fill in behind them (list scheduling).  key = (order - 2) * 64 + (64 - S).
struct TwistCostOrder {
    unsigned char key_at[4 * MIXED_NCLS];
};
// relative time of one work unit per segment at orders 2..5 (tools/twist_unit_time.py: 60 / 90 / 128 / 210 us at 64 segments);
// listing the classes order by order instead measured the same
constexpr int TWIST_STEP_COST[4] = {47, 70, 100, 164};
constexpr TwistCostOrder make_twist_cost_order() {
    TwistCostOrder c{};
    int cost[4 * MIXED_NCLS] = {};
    for (int key = 0; key < 4 * MIXED_NCLS; ++key) {
        cost[key] = TWIST_STEP_COST[key / MIXED_NCLS] * (MIXED_NCLS - key % MIXED_NCLS);
        c.key_at[key] = (unsigned char)key;
    }
    for (int i = 1; i < 4 * MIXED_NCLS; ++i) {   // insertion sort, descending, stable
        const unsigned char k = c.key_at[i];
        int j = i;
        while (j > 0 && cost[c.key_at[j - 1]] < cost[k]) { c.key_at[j] = c.key_at[j - 1]; --j; }
        c.key_at[j] = k;
    }
    return c;
}

}  // namespace csp
