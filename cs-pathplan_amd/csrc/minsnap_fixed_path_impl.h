// minsnap_fixed_path_impl.h -- register-resident kernel WITH the path-deviation penalty
// (reference: math_util/minimum_snap.cpp:347-469 + :511-624), same twisted two-wave mapping as
// minsnap_fixed_impl.h.  One launch does what the reference does in one SolveQPClosedForm call:
//
//   pass A  the unpenalised pre-solve (Q_original: no path term, no zero-velocity term, :349-405),
//           whose polynomial is sampled at t = T*s/16, s = 0..16, per segment; the first sample of
//           maximal squared distance to the chord wins (strict '>', :408-439)           -> t*_k
//   pass B  the penalised solve: per segment a rank-1 term w*h h^T with h = M^-T phi(t*) (the
//           Hermite weights at t*, from the constant table HW) plus the linear term
//           f~ = -2 w L(t*) h, applied UN-halved like the reference (:577-579), plus the
//           zero-velocity term; then recovery, stores and the deviation metric at t* (:594-624).
//
// Penalty algebra: for a row of a free derivative x of a segment, with Hermite
// weight h_x,  the known part of  w*h_x*(h^T d) + f~_x  is  -w*h_x*(P_s + (2*tau - h_endpos)*dP)
// because h_startpos + h_endpos = 1 and L = P_s + tau*dP; it moves to the right-hand side as
// +h_x*g with g = w*(P_s + (2*tau - h_endpos)*dP) per axis.  Everything is evaluated in the
// role's own (possibly time-reversed) frame: the cost terms are frame independent.
#pragma once
#include <type_traits>

#include "minsnap_fixed_impl.h"

namespace csp {
namespace fixedk {

// Per-segment quantities of the penalised system in the role's frame.
template <int O> struct PSeg {
    double ip[2 * O];   // T^-e
    double h[2 * O];    // Hermite weights at t*, including the T^deriv scaling (zero in pass A)
    double g[3];        // w*(P_s + (2 tau - h_endpos) dP) per axis (zero in pass A)
    double dP[3];       // P_end - P_start (local frame)
};

template <int O, bool PEN>
__device__ __forceinline__ void pseg_make(double T, double pw, int tau_idx, const double *l_hw, const double (&Ps)[3],
                                          const double (&Pe)[3], PSeg<O> &s) {
    constexpr int M = 2 * O;
    s.ip[0] = 1.0;
    s.ip[1] = fast_rcp(T);
#pragma unroll
    for (int e = 2; e < M; ++e) s.ip[e] = s.ip[e - 1] * s.ip[1];
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) s.dP[ax] = Pe[ax] - Ps[ax];
    if (PEN) {
        double tp[O];
        tp[0] = 1.0;
#pragma unroll
        for (int e = 1; e < O; ++e) tp[e] = tp[e - 1] * T;
#pragma unroll
        for (int a = 0; a < M; ++a) s.h[a] = l_hw[tau_idx * M + a] * tp[a % O];
        const double k = 2.0 * (double)tau_idx * 0.0625 - s.h[O];
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) s.g[ax] = pw * __builtin_fma(k, s.dP[ax], Ps[ax]);
    } else {
#pragma unroll
        for (int a = 0; a < M; ++a) s.h[a] = 0.0;
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) s.g[ax] = 0.0;
    }
}

// block entries of Qt(T) + w h h^T (free derivative index r <-> endpoint-derivative index r+1).  PEN = false (pass A, the
// unpenalised pre-solve): the plain table entry -- h is zero there, but fma(0, 0, x) is not x to the compiler (signed zeros),
// so the products were really issued.
template <int O, bool PEN> __device__ __forceinline__ double q_ss(const PSeg<O> &s, double pw, int r, int c) {
    const double q = Tab<O>::QT(r + 1, c + 1) * s.ip[2 * O - 3 - r - c];
    return PEN ? __builtin_fma(pw * s.h[r + 1], s.h[c + 1], q) : q;
}
template <int O, bool PEN> __device__ __forceinline__ double q_se(const PSeg<O> &s, double pw, int r, int c) {
    const double q = Tab<O>::QT(r + 1, O + c + 1) * s.ip[2 * O - 3 - r - c];
    return PEN ? __builtin_fma(pw * s.h[r + 1], s.h[O + c + 1], q) : q;
}
template <int O, bool PEN> __device__ __forceinline__ double q_ee(const PSeg<O> &s, double pw, int r, int c) {
    const double q = Tab<O>::QT(O + r + 1, O + c + 1) * s.ip[2 * O - 3 - r - c];
    return PEN ? __builtin_fma(pw * s.h[O + r + 1], s.h[O + c + 1], q) : q;
}

// Dense residency (order 2, S >= 8): under 40 KB of LDS and 256 registers, so that FOUR workgroups share a CU.
template <int O, int S> constexpr bool path_dense = (O == 2 && S >= 8);
// doubles per staging row: the dense variant gives up the two padding doubles (2-way conflicts on its six
// ds_write_b128 per segment, a few dozen clocks) for 2 KB of LDS
template <int O, int S> constexpr int path_stage_row = path_dense<O, S> ? FixedLds<O, S>::REC : FixedLds<O, S>::STAGE_ROW;
// Dense residency WITH whole-line stores (round 3; order 2, S = 8, 12, 16 -- trajectories that start on lines): a
// 64-row ring tile does not fit under 40 KB, so the tile has 32 rows and a segment's records leave in two half-wave
// phases (HalfRing below); the bytes a phase would have to keep in the tile for the next record are kept in REGISTERS
// instead (they are part of the previous record, which the lane still holds) and re-written with it.
template <int O, int S> constexpr bool path_dense_ring = path_dense<O, S> && LineGeom<O, S>::OK;
template <int O, int S> constexpr int path_tile_doubles = path_dense_ring<O, S> ? 32 * 26 : 64 * path_stage_row<O, S>;

// Half-height ring of the dense order-2 kernel: rows of 192 ring bytes (+ 16 bytes of padding: the stride in dwords is an
// odd multiple of 4), 32 rows.  A 128-byte line that starts at ring byte 128 wraps (pieces 4..7 sit at ring bytes 0..63).
template <int S, bool BOTTOM> struct HalfRing {
    static constexpr int RECB = 96, RINGB = 192, ROW = 26, HT = (S + 1) / 2, RS = S * RECB;
    static constexpr int LO = BOTTOM ? HT * RECB : 0, HI = BOTTOM ? S * RECB : HT * RECB;
    __device__ static __forceinline__ constexpr int pos(int x) { return (x % RINGB) / 8; }   // doubles
    // One segment's records of the slice: cn = this lane's new record (12 doubles, [axis][power]), cp = its previous one
    // (segment g+1 for the top role, g-1 for the bottom role; unused at the role's first record).
    __device__ static __forceinline__ void put_and_flush(int g, const double (&cn)[12], const double (&cp)[12], double *tile, char *tbase,
                                                         int lane, bool nt, unsigned live8) {
        const int rlo = g * RECB, rhi = rlo + RECB;
        constexpr int C128 = 128;
        // bytes of the previous record that still wait for their line
        const int hlo = BOTTOM ? ((rlo / C128) * C128 > LO ? (rlo / C128) * C128 : LO) : rhi;
        const int hhi = BOTTOM ? rlo : ((((rhi + 127) / C128) * C128) < HI ? (((rhi + 127) / C128) * C128) : HI);
        const int prlo = BOTTOM ? rlo - RECB : rhi;     // where the previous record starts
        const int l0 = BOTTOM ? rlo / C128 : (rlo + 127) / C128;
        const int nl = BOTTOM ? rhi / C128 - rlo / C128 : (rhi + 127) / C128 - (rlo + 127) / C128;   // 0 or 1
        const int q = lane >> 3, p = lane & 7;
        const int row32 = lane & 31;
        const unsigned g_lane = (unsigned)(q * RS + p * 16);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if ((lane >> 5) == h) {
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    if (prlo + 16 * k >= hlo && prlo + 16 * k + 16 <= hhi) {
                        double2 v2;
                        v2.x = cp[2 * k];
                        v2.y = cp[2 * k + 1];
                        *reinterpret_cast<double2 *>(tile + row32 * ROW + pos(prlo + 16 * k)) = v2;
                    }
                }
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    double2 v2;
                    v2.x = cn[2 * k];
                    v2.y = cn[2 * k + 1];
                    *reinterpret_cast<double2 *>(tile + row32 * ROW + pos(rlo + 16 * k)) = v2;
                }
            }
            // Half the lanes wrote, all lanes read: the exchange crosses lanes THROUGH a divergent branch.  Per thread the
            // reads below do not depend on the (not taken) writes, so without a CONVERGENT marker the compiler may duplicate
            // the tail into both sides of the branch and run the non-writers' reads first (it did: rows 32..35 of every
            // slice read the other half's bytes).  wave_barrier is convergent; the fences pin the memory order around it.
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (nl > 0) {
                const int ell = l0;
                const bool whole = BOTTOM ? ell * 128 >= LO : ell * 128 + 128 <= HI;
                const bool pv = whole || (BOTTOM ? ell * 128 + p * 16 >= LO : ell * 128 + p * 16 < HI);
                const int c = (ell * 128) % RINGB;                        // 0, 64 or 128
                const int poff = c == 128 ? (p < 4 ? 16 + p * 2 : (p - 4) * 2) : c / 8 + p * 2;
                double2 v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const double2 *>(tile + (i * 8 + q) * ROW + poff);
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (pv && ((live8 >> (h * 4 + i)) & 1u)) store16(tbase + ell * 128 + (size_t)(h * 32 + i * 8) * RS + g_lane, v[i], nt);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();   // every lane has read before the other half overwrites the tile
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
};

// Dense residency: a role's segment times live in registers for the whole kernel (waypoints still come from LDS), so
// that the staging tiles can lie on top of the LDS image of the times.
template <int S, bool BOTTOM> struct TimeRegInputs {
    static constexpr int HS = BOTTOM ? S / 2 : (S + 1) / 2;
    double t[HS];
    const double *l_wp;
    int lane;
    __device__ __forceinline__ void fill(const LdsInputs<S, BOTTOM> &in) {
        l_wp = in.l_wp;
        lane = in.lane;
#pragma unroll
        for (int j = 0; j < HS; ++j) t[j] = in.T(j);
    }
    __device__ __forceinline__ double T(int j) const { return t[j]; }
    __device__ __forceinline__ double P(int j, int ax) const { return l_wp[lane * (S + 1) * 3 + (BOTTOM ? S - j : j) * 3 + ax]; }
};

// One full twisted sweep.  PEN = false: pass A (fills tau[]); PEN = true: pass B (stores, deviation).
template <int O, int S, bool BOTTOM, bool PEN, bool STATUS, class In>
__device__ __forceinline__ void path_sweep(const GenericArgs &a, int64_t b0, int rows, int lane, const In &in,
                                           const RoleBc<BOTTOM> &rbc, double pw, const double *l_hw, double *stage,
                                           double *xchg, double *partner_xchg, const int *l_skip,
                                           int (&tau)[BOTTOM ? S / 2 : (S + 1) / 2], bool &spd,
                                           double &nanacc, double &maxdev) {
    constexpr int N = O - 1, M = 2 * O;
    constexpr int HS = BOTTOM ? S / 2 : (S + 1) / 2;
    using L = FixedLds<O, S>;
    const double vw = PEN ? rbc.vw : 0.0;  // the pre-solve uses Q_original (:349)
    const double w = PEN ? pw : 0.0;

    double z[N][3], W[N][N];
#pragma unroll
    for (int r = 0; r < N; ++r) {
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) z[r][ax] = rbc.at(r, ax);
#pragma unroll
        for (int c = 0; c < N; ++c) W[r][c] = 0.0;
    }
    double Wst[HS][N][N], zst[HS][N][3];

    PSeg<O> left, right;
    double Pa[3], Pb[3], Pc[3];
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) { Pa[ax] = in.P(0, ax); Pb[ax] = in.P(1, ax); }
    pseg_make<O, PEN>(in.T(0), w, tau[0], l_hw, Pa, Pb, left);
#pragma unroll
    for (int k = 1; k < HS; ++k) {
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) Pc[ax] = in.P(k + 1, ax);
        pseg_make<O, PEN>(in.T(k), w, tau[k], l_hw, Pb, Pc, right);
        double seL[N][N];  // the left segment's coupling block (used by the matrix AND the right-hand side)
#pragma unroll
        for (int j = 0; j < N; ++j)
#pragma unroll
            for (int r = 0; r < N; ++r) seL[j][r] = q_se<O, PEN>(left, w, j, r);
        double Sm[N][N], R[N][N + 3];
#pragma unroll
        for (int r = 0; r < N; ++r) {
#pragma unroll
            for (int c = 0; c <= r; ++c) {
                double v = q_ee<O, PEN>(left, w, r, c) + q_ss<O, PEN>(right, w, r, c);
                if (r == 0 && c == 0) v += 2.0 * vw;  // both neighbours' velocity terms (:473-509)
#pragma unroll
                for (int j = 0; j < N; ++j) v = __builtin_fma(-seL[j][r], W[j][c], v);
                Sm[r][c] = v;
            }
#pragma unroll
            for (int c = 0; c < N; ++c) R[r][c] = q_se<O, PEN>(right, w, r, c);
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) {
                double v = (Tab<O>::QT(O + r + 1, 0) * left.ip[M - 2 - r]) * left.dP[ax];
                v = __builtin_fma(Tab<O>::QT(r + 1, 0) * right.ip[M - 2 - r], right.dP[ax], v);
                if (PEN) {
                    v = __builtin_fma(left.h[O + r + 1], left.g[ax], v);
                    v = __builtin_fma(right.h[r + 1], right.g[ax], v);
                }
#pragma unroll
                for (int j = 0; j < N; ++j) v = __builtin_fma(-seL[j][r], z[j][ax], v);
                R[r][N + ax] = v;
            }
        }
        spd &= SmallSpd<N, N + 3>::solve(Sm, R);
#pragma unroll
        for (int r = 0; r < N; ++r) {
#pragma unroll
            for (int c = 0; c < N; ++c) { W[r][c] = R[r][c]; Wst[k][r][c] = R[r][c]; }
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) { z[r][ax] = R[r][N + ax]; zst[k][r][ax] = R[r][N + ax]; }
        }
        left = right;
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) { Pa[ax] = Pb[ax]; Pb[ax] = Pc[ax]; }
    }

    // ---- carry onto the middle waypoint, exchange, middle solve ----
    double Cm[N][N], cm[N][3];
    {
        double seL[N][N];
#pragma unroll
        for (int j = 0; j < N; ++j)
#pragma unroll
            for (int r = 0; r < N; ++r) seL[j][r] = q_se<O, PEN>(left, w, j, r);
#pragma unroll
        for (int r = 0; r < N; ++r) {
#pragma unroll
            for (int c = 0; c <= r; ++c) {
                double v = q_ee<O, PEN>(left, w, r, c);
                if (r == 0 && c == 0) v += vw;
#pragma unroll
                for (int j = 0; j < N; ++j) v = __builtin_fma(-seL[j][r], W[j][c], v);
                Cm[r][c] = v;
            }
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) {
                double v = (Tab<O>::QT(O + r + 1, 0) * left.ip[M - 2 - r]) * left.dP[ax];
                if (PEN) v = __builtin_fma(left.h[O + r + 1], left.g[ax], v);
#pragma unroll
                for (int j = 0; j < N; ++j) v = __builtin_fma(-seL[j][r], z[j][ax], v);
                cm[r][ax] = v;
            }
        }
    }
    lds_barrier();  // the partner is done with whatever it last read from / staged in its tile
    {
        int e = 0;
#pragma unroll
        for (int r = 0; r < N; ++r)
#pragma unroll
            for (int c = 0; c <= r; ++c) partner_xchg[(e++) * 64 + lane] = Cm[r][c];
#pragma unroll
        for (int r = 0; r < N; ++r)
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) partner_xchg[(e++) * 64 + lane] = cm[r][ax];
    }
    lds_barrier();
    double xn[N][3];
    {
        double Sm[N][N], R[N][3];
        int e = 0;
#pragma unroll
        for (int r = 0; r < N; ++r)
#pragma unroll
            for (int c = 0; c <= r; ++c) {
                const double o = xchg[(e++) * 64 + lane];
                Sm[r][c] = Cm[r][c] + (((r + c) & 1) ? -o : o);
            }
#pragma unroll
        for (int r = 0; r < N; ++r)
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) {
                const double o = xchg[(e++) * 64 + lane];
                R[r][ax] = cm[r][ax] + ((r & 1) ? o : -o);
            }
        spd &= SmallSpd<N, 3>::solve(Sm, R);
#pragma unroll
        for (int r = 0; r < N; ++r)
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) xn[r][ax] = R[r][ax];
    }

    if (PEN) CSP_STAMP(3);
    // ---- backward sweep ----
    constexpr int RECB = L::REC * 8, RS = S * RECB, ROW = path_stage_row<O, S>;
    const int grp = lane / L::LPR, lane_in = lane - grp * L::LPR;
    const int lds_off = grp * ROW + lane_in * 2;
    const unsigned g_off = (unsigned)(grp * RS + lane_in * 16);
    // whole-line stores through the ring of minsnap_fixed_impl.h (LineRing) wherever the trajectories start on 128-byte
    // lines; the dense order-2 variant keeps its 96-byte tile rows (no room for a ring under 40 KB of LDS)
    constexpr bool RING = PEN && LineGeom<O, S>::OK && !path_dense<O, S>;
    constexpr bool DRING = PEN && path_dense_ring<O, S>;   // order 2 only: half-height ring, records held in registers
    using LR = LineRing<O, S, BOTTOM>;
    double cprev[DRING ? 12 : 1] = {};
    unsigned live8 = 0;   // bit i: row i*8 + lane/8 is a live trajectory of this slice
    if (RING || DRING) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = i * 8 + (lane >> 3);
            live8 |= (r < rows && !l_skip[r]) ? (1u << i) : 0u;
        }
    }
#pragma unroll
    for (int j = HS - 1; j >= 0; --j) {
        double xk[N][3];
#pragma unroll
        for (int r = 0; r < N; ++r)
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) {
                if (j == 0) {
                    xk[r][ax] = rbc.at(r, ax);
                } else {
                    double v = zst[j][r][ax];
#pragma unroll
                    for (int c = 0; c < N; ++c) v = __builtin_fma(-Wst[j][r][c], xn[c][ax], v);
                    xk[r][ax] = v;
                }
            }
        const double Tj = in.T(j);
        double ip[M], tp[N];
        ip[0] = 1.0;
        ip[1] = fast_rcp(Tj);
#pragma unroll
        for (int e = 2; e < M; ++e) ip[e] = ip[e - 1] * ip[1];
        tp[0] = Tj;
#pragma unroll
        for (int e = 1; e < N; ++e) tp[e] = tp[e - 1] * Tj;
        double P0[3], P1[3];
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) { P0[ax] = in.P(j, ax); P1[ax] = in.P(j + 1, ax); }
        if (!PEN) {
            // pass A: where does the pre-solve polynomial stray farthest from the chord?  Samples in
            // GLOBAL order 0..16 (the bottom role's frame is reversed), first maximum wins (strict >, :435).
            // The deviation e(sigma) = P(sigma) - L(sigma) vanishes at both waypoints, so per axis
            //   e(sigma) = sigma (1 - sigma) q(u),  u = sigma - 1/2,  q of degree 2o-3 (table KQ, tablegen.py),
            // and d2(s) = (sigma (1 - sigma))^2 |q(u_s)|^2.  Samples s and 16 - s are +-u: q(+-u) = E(u^2) +- u O(u^2),
            // one even/odd Horner pass for both; samples 0 and 16 are exactly 0 (the reference's are rounding noise
            // there: its search starts from sample 0 as well and never ends on sample 16).
            constexpr int NQ = M - 2;   // coefficients of q
            double cq[3][NQ];
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) {
                const double dp = P1[ax] - P0[ax];
                double hs[N], he[N];
#pragma unroll
                for (int r = 0; r < N; ++r) { hs[r] = xk[r][ax] * tp[r]; he[r] = xn[r][ax] * tp[r]; }
#pragma unroll
                for (int i = 0; i < NQ; ++i) {
                    double v = Tab<O>::KQ(0, i) * dp;
#pragma unroll
                    for (int r = 0; r < N; ++r) {
                        v = __builtin_fma(Tab<O>::KQ(1 + r, i), hs[r], v);
                        v = __builtin_fma(Tab<O>::KQ(1 + N + r, i), he[r], v);
                    }
                    cq[ax][i] = v;
                }
            }
            // Strict '>' scan in GLOBAL sample order 0..16 = the first maximum.  Sample 0 is the starting value (d2 = 0);
            // a pair (u = -+p/16) yields global samples 8-p and 8+p.  The low samples arrive in scan order (p = 7..1:
            // globals 1..7) and take a strict '>'; the high ones arrive in REVERSE scan order (globals 15..9), where '>='
            // keeps the smallest index among equals; the two halves and the middle sample are then combined in scan
            // order.  A rolled loop on purpose: unrolled, the seven pairs' temporaries compete with the stored factors
            // (most of the register file) and the kernel spills.
            double best_lo = 0.0, best_hi = -1.0;
            int g_lo = 0, g_hi = 16;                  // global sample indices
            if constexpr (O == 2) {
                // Order 2: q is LINEAR in u per axis, so |q(u)|^2 = A + u Bc + u^2 C with three scalars per segment
                // (A = |q0|^2, Bc = 2 q0.q1, C = |q1|^2): 5 operations per sample pair instead of ~20, and the pairs
                // unrolled (u, u^2 and the weights are literals).  The search was 13 k of a wave's 57 k clocks -- and
                // what a workgroup does before its first store is what the whole launch waits for (round 3 stamps).
                const double A = __builtin_fma(cq[2][0], cq[2][0], __builtin_fma(cq[1][0], cq[1][0], cq[0][0] * cq[0][0]));
                const double C = __builtin_fma(cq[2][1], cq[2][1], __builtin_fma(cq[1][1], cq[1][1], cq[0][1] * cq[0][1]));
                const double Bc = 2.0 * __builtin_fma(cq[2][0], cq[2][1], __builtin_fma(cq[1][0], cq[1][1], cq[0][0] * cq[0][1]));
#pragma unroll
                for (int p_ = 7; p_ >= 1; --p_) {
                    const double u = (double)p_ * 0.0625, u2 = u * u;
                    const double hw = 0.25 - u2, wgt = hw * hw;
                    const double t = __builtin_fma(u2, C, A);
                    const double sp = __builtin_fma(u, Bc, t), sm = __builtin_fma(-u, Bc, t);
                    const double lo = wgt * (BOTTOM ? sp : sm), hi = wgt * (BOTTOM ? sm : sp);
                    if (lo > best_lo) { best_lo = lo; g_lo = 8 - p_; }
                    if (hi >= best_hi) { best_hi = hi; g_hi = 8 + p_; }
                }
            } else {
#pragma unroll 1
            for (int p_ = 7; p_ >= 1; --p_) {
                const double u = (double)p_ * 0.0625, u2 = u * u;
                const double hw = 0.25 - u2, wgt = hw * hw;
                double sp = 0.0, sm = 0.0;
#pragma unroll
                for (int ax = 0; ax < 3; ++ax) {
                    // even part: coefficients 0, 2, ..; odd part: 1, 3, ..  (NQ is even: NQ/2 of each)
                    double ev = cq[ax][NQ - 2], od = cq[ax][NQ - 1];
#pragma unroll
                    for (int i = NQ - 4; i >= 0; i -= 2) {
                        ev = __builtin_fma(ev, u2, cq[ax][i]);
                        od = __builtin_fma(od, u2, cq[ax][i + 1]);
                    }
                    const double qp = __builtin_fma(od, u, ev), qm = __builtin_fma(-od, u, ev);
                    sp = __builtin_fma(qp, qp, sp);
                    sm = __builtin_fma(qm, qm, sm);
                }
                // local samples 8 - p (u < 0: sm) and 8 + p (u > 0: sp); global index = local (top role) or 16 - local (bottom)
                const double lo = wgt * (BOTTOM ? sp : sm), hi = wgt * (BOTTOM ? sm : sp);
                if (lo > best_lo) { best_lo = lo; g_lo = 8 - p_; }
                if (hi >= best_hi) { best_hi = hi; g_hi = 8 + p_; }
            }
            }
            double best = best_lo;
            int best_g = g_lo;
            {
                const double mid = 0.0625 * __builtin_fma(cq[2][0], cq[2][0], __builtin_fma(cq[1][0], cq[1][0], cq[0][0] * cq[0][0]));
                if (mid > best) { best = mid; best_g = 8; }
            }
            if (best_hi > best) { best = best_hi; best_g = g_hi; }
            const int best_s = BOTTOM ? 16 - best_g : best_g;   // local sample index
            tau[j] = best_s;
        } else {
            const int g = BOTTOM ? S - 1 - j : j;
            double d2 = 0.0, len2 = 0.0;
            double cnew[DRING ? 12 : 1];
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) {
                double xs[N], xe[N], c[M];
#pragma unroll
                for (int r = 0; r < N; ++r) {
                    const double sgn = (BOTTOM && !(r & 1)) ? -1.0 : 1.0;
                    xs[r] = BOTTOM ? sgn * xn[r][ax] : xk[r][ax];
                    xe[r] = BOTTOM ? sgn * xk[r][ax] : xn[r][ax];
                }
                const double Ps = BOTTOM ? P1[ax] : P0[ax], Pe = BOTTOM ? P0[ax] : P1[ax];
                recover<O>(Ps, Pe - Ps, xs, xe, tp, ip, c);
                if constexpr (DRING) {
#pragma unroll
                    for (int i = 0; i < M; ++i) cnew[ax * M + i] = c[i];
                } else {
#pragma unroll
                    for (int i = 0; i < M; i += 2) {
                        double2 v2;
                        v2.x = c[i];
                        v2.y = c[i + 1];
                        const int at = RING ? LR::pos(g * RECB + (ax * M + i) * 8) : ax * M + i;
                        *reinterpret_cast<double2 *>(stage + lane * ROW + at) = v2;
                    }
                }
                if (STATUS) {
                    // Non-finite values are caught on the highest-power and the constant coefficient: every endpoint
                    // quantity of the segment (dP, the 2(o-1) scaled derivatives) enters c[0] with a non-zero Hermite
                    // weight and T^-(2o-1), and c[M-1] is the start waypoint, so a NaN/Inf anywhere in the inputs or
                    // the unknowns reaches one of the two.  Testing all 2o coefficients (as the plain kernel does)
                    // keeps them live across the staging writes: 0.6 KB of scratch per lane at order 4, S >= 13.
                    nanacc = __builtin_fma(c[0], 0.0, nanacc);
                    nanacc = __builtin_fma(c[M - 1], 0.0, nanacc);
                }
                // deviation at the recorded t* (:596-617), in the local frame: P(t*) = h^T d
                double v = __builtin_fma(l_hw[tau[j] * M + O], P1[ax], l_hw[tau[j] * M] * P0[ax]);
#pragma unroll
                for (int r = 0; r < N; ++r) {
                    v = __builtin_fma(l_hw[tau[j] * M + r + 1] * tp[r], xk[r][ax], v);
                    v = __builtin_fma(l_hw[tau[j] * M + O + r + 1] * tp[r], xn[r][ax], v);
                }
                const double dp = P1[ax] - P0[ax];
                const double Lc = __builtin_fma((double)tau[j] * 0.0625, dp, P0[ax]);
                d2 = __builtin_fma(v - Lc, v - Lc, d2);
                len2 = __builtin_fma(dp, dp, len2);
            }
            {
                // deviation / chord length (:612-616: chords of <= 1e-6 are skipped).
                // Order 2: `maxdev` carries the SQUARED ratio (one square root per trajectory, in path_role: two square
                // roots and a division per segment were a fifth of this short sweep).
                // Whether the test is a REAL branch (the empty asm keeps it one) decides the register allocation: a
                // branch ends the basic block once per segment; without it the unrolled segments of a half form ONE
                // scheduling region.
                //  * order 2, S < 8: branch-free -- registers are plentiful and the interleaving hides the fp64
                //    latencies of a lone wave (46.8 us against 54.2 with a branch at one wave per SIMD);
                //  * order 2, S >= 8 (path_dense): branch -- 169 registers instead of 442, which is what lets two
                //    waves share a SIMD (see the kernel);
                //  * orders 3-4: branch -- as one block the segments are register-allocated together and, with the
                //    stored factors filling most of the file, end in scratch (1.4 KB/lane at order 4, S = 16: 170 us
                //    instead of 98).
                double ratio = 0.0;
                if constexpr (O == 2 && !path_dense<O, S>) {
                    ratio = (len2 > 1e-12) ? d2 * fast_rcp(len2) : 0.0;
                } else {
                    if (len2 > 1e-12) {
                        asm volatile("" ::: "memory");
                        ratio = O == 2 ? d2 * fast_rcp(len2) : sqrt(d2 * fast_rcp(len2));
                    }
                }
                maxdev = ratio > maxdev ? ratio : maxdev;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            if constexpr (DRING) {
                HalfRing<S, BOTTOM>::put_and_flush(g, cnew, cprev, stage, reinterpret_cast<char *>((double *)a.coeffs + b0 * S * L::REC), lane,
                                                   a.nt_stores != 0, live8);
#pragma unroll
                for (int i = 0; i < 12; ++i) cprev[i] = cnew[i];
            } else if (RING) {
                LR::template flush<true>(g, stage, reinterpret_cast<char *>((double *)a.coeffs + b0 * S * L::REC), lane, a.nt_stores != 0, live8);
            } else {
                char *gbase = reinterpret_cast<char *>((double *)a.coeffs + (b0 * S + g) * L::REC);
#pragma unroll
                for (int i = 0; i < L::NI; ++i) {
                    const int row = i * L::RPI + grp;
                    if (lane < L::RPI * L::LPR && row < rows && !l_skip[row]) {
                        const double2 v2 = *reinterpret_cast<const double2 *>(stage + lds_off + i * L::RPI * ROW);
                        store16(gbase + (size_t)i * L::RPI * RS + g_off, v2, a.nt_stores != 0);
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        }
#pragma unroll
        for (int r = 0; r < N; ++r)
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) xn[r][ax] = xk[r][ax];
    }
}

template <int O, int S, bool BOTTOM, bool STATUS, bool DENSE>
__device__ __forceinline__ void path_role(const GenericArgs &a, int64_t b0, int rows, int64_t b, int lane,
                                          const double *l_wp, const double *l_tm, const double *l_hw, double *stage,
                                          double *xchg, double *partner_xchg, const int *l_skip, double *l_dev, int *l_bits) {
    constexpr int HS = BOTTOM ? S / 2 : (S + 1) / 2;
    const LdsInputs<S, BOTTOM> lin{l_wp, l_tm, lane};
    // DENSE: the staging tiles lie on the LDS image of the times; the first write into a tile (pass A's exchange, or
    // pass B's with tau_mode 2) follows an lds_barrier that the partner reaches only after its own fill.
    typename std::conditional<DENSE, TimeRegInputs<S, BOTTOM>, LdsInputs<S, BOTTOM>>::type in;
    if constexpr (DENSE) in.fill(lin);
    else in = lin;
    RoleBc<BOTTOM> rbc;
    rbc.load(a, b);
    int tau[HS];
#pragma unroll
    for (int j = 0; j < HS; ++j) tau[j] = 0;
    bool spd = true;
    double nanacc = 0.0, maxdev = 0.0;
    // The pre-solve uses Q_original (no path term, no zero-velocity term, :349), so inside the re-solve
    // loop -- only vel_zero_weight changes between its passes (:80-90) -- every pass finds the same t*.
    // tau_mode 2 reads the sample indices a previous pass stored (global order, [segment][trajectory])
    // instead of repeating the pre-solve and the 17-sample search; tau_mode 1 stores them.
    if (a.tau_mode == 2) {
#pragma unroll
        for (int j = 0; j < HS; ++j) {
            const int sg = a.tstar[(int64_t)(BOTTOM ? S - 1 - j : j) * a.B + b];
            tau[j] = BOTTOM ? 16 - sg : sg;
        }
    } else {
        path_sweep<O, S, BOTTOM, false, STATUS>(a, b0, rows, lane, in, rbc, a.path_weight, l_hw, stage, xchg, partner_xchg, l_skip, tau, spd, nanacc, maxdev);
        if (a.tau_mode == 1 && lane < rows) {
#pragma unroll
            for (int j = 0; j < HS; ++j) a.tstar[(int64_t)(BOTTOM ? S - 1 - j : j) * a.B + b] = BOTTOM ? 16 - tau[j] : tau[j];
        }
    }
    spd = true;  // the reported status is the penalised solve's
    CSP_STAMP(2);
    path_sweep<O, S, BOTTOM, true, STATUS>(a, b0, rows, lane, in, rbc, a.path_weight, l_hw, stage, xchg, partner_xchg, l_skip, tau, spd, nanacc, maxdev);
    // the reference's max_deviation is the maximum over ALL segments and the status covers both
    // halves: the bottom role hands its part to the top role, which writes (plain stores, and only
    // for live trajectories: a re-solve pass must leave finished trajectories untouched)
    const int bits = (spd ? 0 : 2) | ((nanacc == 0.0) ? 0 : 1);
    CSP_STAMP(4);
    CSP_STAMP_RT(6);
    lds_barrier();
    if (BOTTOM) { l_dev[lane] = maxdev; l_bits[lane] = bits; }
    lds_barrier();
    const bool live = lane < rows && !l_skip[lane];
    if (!BOTTOM && live) {
        if (a.max_dev) a.max_dev[b] = O == 2 ? sqrt(fmax(maxdev, l_dev[lane])) : fmax(maxdev, l_dev[lane]);
        if (STATUS) a.status[b] = bits | l_bits[lane];
    }
}

// One workgroup per 64-trajectory slice (path-penalty solves are not the streaming headline: no
// persistent/prefetch structure here).  `skip` marks trajectories the re-solve loop has finished.
template <int O, int S, bool STATUS>
__global__ void __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(path_dense<O, S> ? 2 : 1)))
minsnap_fixed_path_kernel(GenericArgs a) {
    using L = FixedLds<O, S>;
    constexpr int M = 2 * O;
    // DENSE (order 2, S >= 8): the staging tiles lie on top of the LDS image of the segment times (each role keeps its
    // times in registers) and lose their padding, which brings the workgroup under 40 KB of LDS: four workgroups per
    // CU = two waves per SIMD (the order-2 sweep fits 256 registers once its segments are separate basic blocks),
    // and the 1024 workgroups of a 65536-trajectory batch are resident at once.
    constexpr bool DENSE = path_dense<O, S>;
    constexpr int TILE = path_tile_doubles<O, S>;
    static_assert(L::CARRY * 64 <= TILE, "carries must fit in a staging tile");
    static_assert(!DENSE || 64 * S <= 2 * TILE, "the image of the times must fit under the staging tiles");
    // DENSE: the end-of-kernel hand-over of max_dev / status bits (l_dev, l_bits) lies on the top role's tile, which is
    // dead by then (both waves have passed a barrier after their last staged store)
    static_assert(!DENSE || 64 + 32 <= TILE, "l_dev + l_bits must fit in a staging tile");
    __shared__ __attribute__((aligned(16))) double lds[L::WP_DOUBLES + (DENSE ? 0 : 64 * S) + 2 * TILE + 17 * M + (DENSE ? 0 : 64) + (DENSE ? 32 : 64)];
    double *l_wp = lds;
    double *l_tm = l_wp + L::WP_DOUBLES;
    double *l_stage = DENSE ? l_tm : l_tm + 64 * S;
    double *l_hw = l_stage + 2 * TILE;
    double *l_dev = DENSE ? l_stage : l_hw + 17 * M;
    int *l_skip = reinterpret_cast<int *>(DENSE ? l_hw + 17 * M : l_dev + 64);
    int *l_bits = DENSE ? reinterpret_cast<int *>(l_stage + 64) : l_skip + 64;
    const int tid = threadIdx.x, lane = tid & 63, role = tid >> 6;
    const int64_t b0 = (int64_t)blockIdx.x * 64;
    const int rows = (int)((a.B - b0) < 64 ? (a.B - b0) : 64);
    if (a.skip) {
        // re-solve loop: a slice whose trajectories have all converged costs one load (both waves see
        // the same mask, so both leave)
        const int sk = lane < rows ? a.skip[b0 + lane] : 1;
        if (__builtin_amdgcn_ballot_w64(sk == 0) == 0) return;
    }
    CSP_STAMP_RT(5);
    CSP_STAMP(0);
    if constexpr (!DENSE) {
        // Orders 3-4 (and short order-2 trajectories): two workgroups per CU, one wave per SIMD.  Started together, the two
        // compute together and then store together; the second workgroup of a CU (dispatch order: blocks 256..511 of the
        // first round) starts a.stagger x 8128 clocks late so that one's store phase meets the other's compute.
        if (a.stagger > 0 && gridDim.x > 256 && blockIdx.x >= 256 && blockIdx.x < 512)
            for (int q = 0; q < a.stagger; ++q) __builtin_amdgcn_s_sleep(127);
    }
    if constexpr (DENSE) {
        // A workgroup stores nothing before its backward sweep: started together, the resident workgroups compute with
        // the memory system idle and then all store at once (time = compute-before-the-first-store + bytes / bandwidth:
        // 40 us at B = 65536, S = 16).  The four workgroups of a CU (dispatch order: 256 CUs per round) start 6 S x 64
        // clocks apart instead, so that the store phase of one overlaps the compute of the next: 37.2 us (steps of
        // 0 / 2 S / 4 S / 6 S / 8 S x 64 clocks: 40.1 / 39.9 / 37.7 / 37.2 / 38.2 us).  First round only -- later rounds
        // are out of step by themselves -- and never for grids of <= 256 workgroups.
        const int slot = blockIdx.x < 1024 ? (int)(blockIdx.x >> 8) : 0;
        for (int q = 0; q < slot * a.stagger; ++q) __builtin_amdgcn_s_sleep(S < 127 ? S : 127);   // a.stagger x S x 64 clocks per slot
    }
    {
        const double2 *g_wp = reinterpret_cast<const double2 *>((const double *)a.wp + b0 * L::WP_ROW);
        const int n_wp = rows * L::WP_ROW / 2;
        constexpr int WP_ITERS = (64 * L::WP_ROW / 2 + 127) / 128;
        // Branch-free: pieces beyond the slice are clamped to its last piece (re-writing that piece with its own value)
        // instead of being skipped.  With a conditional per piece every load sat in its own basic block and hipcc
        // emitted load -> s_waitcnt vmcnt(0) -> ds_write thirteen times in a row: thirteen exposed memory round trips
        // at the start of every workgroup.  As one block the loads are all in flight before the first wait.
        {
            double2 v[WP_ITERS];
#pragma unroll
            for (int it = 0; it < WP_ITERS; ++it) {
                const int c = it * 128 + tid;
                v[it] = g_wp[c < n_wp ? c : n_wp - 1];
            }
#pragma unroll
            for (int it = 0; it < WP_ITERS; ++it) {
                const int c = it * 128 + tid;
                reinterpret_cast<double2 *>(l_wp)[c < n_wp ? c : n_wp - 1] = v[it];
            }
        }
        if (((rows * L::WP_ROW) & 1) && tid == 0) l_wp[rows * L::WP_ROW - 1] = ((const double *)a.wp + b0 * L::WP_ROW)[rows * L::WP_ROW - 1];
        const double2 *g_tm = reinterpret_cast<const double2 *>((const double *)a.times + b0 * S);
        const int n_tm = rows * S / 2;
        constexpr int TM_ITERS = (64 * S / 2 + 127) / 128;
        {
            double2 v[TM_ITERS];
#pragma unroll
            for (int it = 0; it < TM_ITERS; ++it) {
                const int c = it * 128 + tid;
                v[it] = g_tm[c < n_tm ? c : n_tm - 1];
            }
#pragma unroll
            for (int it = 0; it < TM_ITERS; ++it) {
                const int c = it * 128 + tid;
                reinterpret_cast<double2 *>(l_tm)[c < n_tm ? c : n_tm - 1] = v[it];
            }
        }
        if (((rows * S) & 1) && tid == 0) l_tm[rows * S - 1] = ((const double *)a.times + b0 * S)[rows * S - 1];
        for (int q = tid; q < 17 * M; q += 128) l_hw[q] = Tab<O>::HW(q / M, q % M);
        if (tid < 64) l_skip[tid] = (tid < rows && a.skip) ? a.skip[b0 + tid] : 0;
    }
    __syncthreads();
    CSP_STAMP(1);
    int64_t b = b0 + lane;
    if (b >= a.B) b = a.B - 1;
    if (role == 0) path_role<O, S, false, STATUS, DENSE>(a, b0, rows, b, lane, l_wp, l_tm, l_hw, l_stage, l_stage, l_stage + TILE, l_skip, l_dev, l_bits);
    else path_role<O, S, true, STATUS, DENSE>(a, b0, rows, b, lane, l_wp, l_tm, l_hw, l_stage + TILE, l_stage + TILE, l_stage, l_skip, l_dev, l_bits);
}

template <int O, int S> hipError_t launch_path_s(const GenericArgs &a, hipStream_t st) {
    const dim3 grid((unsigned)((a.B + 63) / 64)), block(128);
    GenericArgs f = a;
    // Non-temporal coefficient stores (store16) at order 4 from 48 MB of coefficients on (CSP_NT_STORES=0 / 1 forces the
    // choice for A/B runs).  Measured at S = 16 (tools/path_nt_ab.py; three boxes): B = 16384 / 32768 / 65536 / 98304 / 131072 /
    // 262144 / 524288: 40.7 / 45.1 / 93-95 / 145 / 188 / 368 / 723 us against 44.1 / 49.3 / 95.5-97.6 / 163 / 214 / 419 /
    // 817 us with ordinary stores, and WRITE_SIZE 1.013x the coefficients instead of 1.083x: unlike the unpenalised
    // kernel (section 5.1: ordinary stores win while the coefficients fit the Infinity Cache) this one computes for most
    // of its time and gains from lines that leave whole.  Orders 2 and 3 LOSE at every size (B = 524288: 503 against
    // 306 us, 876 against 620 us): their 96- and 144-byte records leave 128-byte lines shared between store
    // instructions, and a non-temporal partial line is not merged on the way out.
    // Launches that write less than 48 MB keep ordinary stores: nothing measurable to gain (B = 8192: 40.2 against
    // 40.6 us; one order-4 flight through the host C-ABI: 1167-1186 us either way), and the reader that follows a small
    // solve (the sampler) then finds the coefficients in the cache.
    const bool big = (double)a.B * S * 6 * O * 8.0 >= 48.0 * 1024 * 1024;
    // Round 3: orders 2 / 3 store whole lines too (LineRing) where the trajectories start on lines -- except the dense
    // order-2 variant -- and take the same rule.
    constexpr bool WHOLE_LINES = O == 4 || LineGeom<O, S>::OK;   // incl. the dense order-2 variant (HalfRing)
    f.nt_stores = nt_forced() >= 0 ? nt_forced() : (WHOLE_LINES && big ? 1 : 0);
    static const int stagger_env = [] { const char *e = std::getenv("CSP_PATH_STAGGER"); return e ? std::atoi(e) : -1; }();
    // measured at B = 65536, S = 16 (tools/path_bench.py, CSP_PATH_STAGGER = 0 / 1 / 2 / 3 / 4 / 6): order 3 58.4 / 57.3 / 57.7 /
    // 61.0 / 65.5 / 76.7 us, order 4 86.4 / 85.0 / 86.8 / 84.0 / 86.2 / 95.7 us -- a 2-3 % effect; scaled with the sweep length
    // (dense order-2 variant: the four workgroups of a CU start stagger x S x 64 clocks apart; with the closed-form search of
    // round 3 -- CSP_PATH_STAGGER = 0 / 2 / 3 / 4 / 5 / 6 / 8: 33.2 / 30.6 / 30.2 / 30.3 / 30.3 / 30.8 / 32.1 us at B = 65536, S = 16)
    f.stagger = stagger_env >= 0 ? stagger_env : (path_dense<O, S> ? 4 : O == 3 ? (S + 8) / 16 : O == 4 ? (3 * S + 8) / 16 : 0);
    if (a.status) hipLaunchKernelGGL((minsnap_fixed_path_kernel<O, S, true>), grid, block, 0, st, f);
    else hipLaunchKernelGGL((minsnap_fixed_path_kernel<O, S, false>), grid, block, 0, st, f);
    return hipGetLastError();
}

}  // namespace fixedk
}  // namespace csp
