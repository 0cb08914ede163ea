// minsnap_twist_launch.h -- what minsnap_mixed.hip needs to know of the lane-pair sweep (minsnap_twist_impl.h).
#pragma once
#include "minsnap_launch.h"
#include "minsnap_mixed.h"

namespace csp {
namespace twist {

constexpr int SMAX = 64;          // longest trajectory of this family
constexpr int HSMAX = SMAX / 2;   // segments per role

template <int O> struct Geo {
    static constexpr int N = O - 1, M = 2 * O;
    // Segments per block, measured (tools/twist_prof.sh, B = 65536 of one order, kernel time; DESIGN.md 10.3b has the table):
    // order 3: 8 of 4 / 8 / 12; order 4: 6 of 2 / 4 / 6 / 8; order 5: 4 of 2 / 3 / 4 / 6.  A small K pays a block start (checkpoint +
    // inputs: one exposed memory round trip for a lone wave) every K segments; a large one pushes the factors K * (N*N + 3N)
    // doubles beyond the 256 architectural registers into AGPR copies and scratch -- and in ONE kernel for all orders the
    // spills of one order slow the others down as well.
#ifndef CSP_TWIST_K5
#define CSP_TWIST_K5 4
#endif
#ifndef CSP_TWIST_K4
#define CSP_TWIST_K4 6
#endif
#ifndef CSP_TWIST_K3
#define CSP_TWIST_K3 8
#endif
    static constexpr int K = O <= 2 ? 8 : O == 3 ? CSP_TWIST_K3 : O == 4 ? CSP_TWIST_K4 : CSP_TWIST_K5;   // segments per block (register budget: K * (N*N + 3N) doubles of factors)
    static constexpr int NCK = (HSMAX + K - 1) / K - 1;      // checkpoints per role (start of blocks 1 ..)
    static constexpr int CKD = N * N + 3 * N;                // doubles per checkpoint and lane: W, z
    static constexpr int CARRY = N * (N + 1) / 2 + 3 * N;    // the Schur carry onto the middle waypoint
    static constexpr size_t CK_DOUBLES_PER_ROLE = (size_t)NCK * CKD * 64;
};

// ONE launch for every order's classes (minsnap_twist_impl.h).  workgroups: the persistent grid; ckws: workgroups * 2 *
// ck_role_doubles doubles of checkpoint slots (ck_role_doubles >= Geo<O>::CK_DOUBLES_PER_ROLE of every order)
hipError_t launch_twist(const GenericArgs &a, bool f32, const int32_t *perm, const int64_t *coef_off, MixedTable *tab, double *ckws,
                        size_t ck_role_doubles, int workgroups, hipStream_t st);

}  // namespace twist
}  // namespace csp
