// twist_order_check.cpp -- CPU check of the work list of csp_minsnap_solve_mixed's lane-pair sweep
// (cs-pathplan_amd/csrc/minsnap_mixed.h: TwistCostOrder, built at compile time; minsnap_twist_launch.h: block geometry).
// Prints the (order, segments) classes in list order, one per line: "pos order S cost".  tests/test_twist_order.py checks: a
// permutation of all 256 classes, cost descending, ties in key order.
#include <cstdio>

#include "../csrc/minsnap_mixed.h"

int main() {
    constexpr csp::TwistCostOrder c = csp::make_twist_cost_order();
    static_assert(sizeof(c.key_at) == 4 * csp::MIXED_NCLS, "one position per (order, S) class");
    for (int p = 0; p < 4 * csp::MIXED_NCLS; ++p) {
        const int key = c.key_at[p], o = key / csp::MIXED_NCLS + 2, S = csp::MIXED_NCLS - key % csp::MIXED_NCLS;
        std::printf("%d %d %d %d\n", p, o, S, csp::TWIST_STEP_COST[o - 2] * S);
    }
    return 0;
}
