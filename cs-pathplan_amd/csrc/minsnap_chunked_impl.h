// minsnap_chunked_impl.h -- device code of the workspace-free kernel for ragged and long trajectories (any S <= 256,
// orders 2..5, fp64 or fp32 storage with fp64 arithmetic, zero-velocity penalty, no path penalty).
//
// The generic kernel (minsnap_generic.hip) walks a trajectory with ONE lane and parks the factors
// W_k, z_k of the block-tridiagonal R_PP (minimum_snap.cpp:564-566) in an HBM workspace -- 2.4x the
// algorithmic bytes at order 4 -- and a batch of 65536 long trajectories is one wave per SIMD.  Here a
// trajectory is cut into nch <= LPT chunks of at most 4 segments and LPT lanes of one wave share it
// (substructuring / nested dissection of the block-tridiagonal system):
//
//   1. each lane eliminates the interior waypoints of ITS chunk twice, left-to-right and (on the
//      time-reversed chunk: reversed waypoints, odd derivatives negated) right-to-left, carrying only
//      the current factor: that yields the chunk's Schur complement onto its two interface waypoints
//      (D_L, D_R symmetric, coupling E, right-hand sides r_L, r_R) without storing anything;
//   2. the interface system -- block-tridiagonal again, nch-1 unknown waypoints -- sits in LDS
//      (24 doubles per interface at order 4).  Lane j solves interface j+1 by a twisted elimination:
//      Schur carries from the left end up to interface j and from the right end down to j+2, nch-2
//      block steps for every lane, so the loop is wave-uniform and needs no storage either; its left
//      interface comes from lane j-1 through LDS;
//   3. with all derivatives known at both ends the chunk is an independent little trajectory: a
//      forward sweep keeping W_k, z_k in registers (<= 3 interior waypoints), back-substitution,
//      Hermite -> monomial recovery (minimum_snap.cpp:582-591) and the stores.
//
// HBM traffic is the algorithmic minimum (inputs once -- the re-reads of steps 1/3 hit L1/L2 --,
// coefficients once); the price is ~3x the arithmetic of the sequential sweep, which one wave per 4..64
// trajectories instead of one lane per trajectory more than pays for.
#pragma once
#include "minsnap_iface.h"

#include <type_traits>

namespace csp {
namespace chunked {

using fixedk::Seg;
using fixedk::SmallSpd;
using fixedk::ee_of;
using fixedk::seg_make;

constexpr int CMAX = 4;  // segments per chunk

template <typename IO> __device__ __forceinline__ double ld(const IO *p) { return (double)*p; }

// This lane's chunk: c segments starting at segment s0 of a trajectory whose first segment is seg0
// (global, ragged prefix) and whose first waypoint is point seg0 + b.
// The chunk's inputs are read ONCE, raw (storage type, forward order), and kept in registers (19 values: 19 VGPRs for fp32
// storage, 38 for fp64): the three passes over the chunk (reversed Schur sweep, forward Schur sweep, solve) used to load them
// three times, and at two waves per SIMD each of those load phases was an exposed memory round trip (round 3).
// Inside an active lane every load is issued unconditionally (indices clamped into the chunk, the unused slots overwritten
// by a select afterwards): with `(i < c) ? load : constant` hipcc put every load into a basic block of its own and
// followed it with s_waitcnt vmcnt(0) -- 19 exposed memory round trips per call, which was more than half of a wave's life.
template <typename IO> struct RawChunk {
    IO t[CMAX], p[CMAX + 1][3];
};

template <typename IO>
__device__ __forceinline__ void load_raw(const IO *wp, const IO *tm, int64_t pt0, int64_t sg0, int c, RawChunk<IO> &r) {
#pragma unroll
    for (int i = 0; i < CMAX; ++i) {
        r.t[i] = (IO)1;
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) r.p[i][ax] = (IO)0;
    }
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) r.p[CMAX][ax] = (IO)0;
    if (c > 0) {   // ONE branch per call (idle lanes touch no memory), none per load
#pragma unroll
        for (int i = 0; i < CMAX; ++i) r.t[i] = tm[sg0 + (i < c ? i : c - 1)];
#pragma unroll
        for (int i = 0; i <= CMAX; ++i) {
            const int64_t q = pt0 + (i <= c ? i : c);
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) r.p[i][ax] = wp[q * 3 + ax];
        }
    }
}

// The chunk in local order from the raw registers; REV: time-reversed (segment i <- c-1-i, waypoint i <- c-i; selects on the
// raw values, c differs between lanes).  Slots beyond the chunk read T = 1, P = 0.
template <typename IO, bool REV>
__device__ __forceinline__ void chunk_view(const RawChunk<IO> &r, int c, double (&T)[CMAX], double (&P)[CMAX + 1][3]) {
#pragma unroll
    for (int i = 0; i < CMAX; ++i) {
        IO v = r.t[i];
        if (REV) {
            v = r.t[0];
#pragma unroll
            for (int k = 1; k < CMAX; ++k) v = (c - 1 - i == k) ? r.t[k] : v;
        }
        T[i] = (i < c) ? (double)v : 1.0;
    }
#pragma unroll
    for (int i = 0; i <= CMAX; ++i)
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
            IO v = r.p[i][ax];
            if (REV) {
                v = r.p[0][ax];
#pragma unroll
                for (int k = 1; k <= CMAX; ++k) v = (c - i == k) ? r.p[k][ax] : v;
            }
            P[i][ax] = (i <= c) ? (double)v : 0.0;
        }
}

// Step 1.  Eliminates the chunk's interior waypoints in local order with the START interface x_s as a
// parameter: x_k = z_k - W_k x_{k+1} - V_k x_s.  Returns the chunk's part of the END interface row
//   D x_e + X x_s = r        (X only when CROSS; it is E^T for the forward direction).
template <int O, bool CROSS>
__device__ __forceinline__ bool chunk_schur(const double (&T)[CMAX], const double (&P)[CMAX + 1][3], int c, double vw,
                                            double (&D)[O - 1][O - 1], double (&rr)[O - 1][3], double (&X)[O - 1][O - 1]) {
    constexpr int N = O - 1, NC = N + 3 + (CROSS ? N : 0);
    double W[N][N], z[N][3], V[N][N];
#pragma unroll
    for (int r = 0; r < N; ++r) {
#pragma unroll
        for (int q = 0; q < N; ++q) { W[r][q] = 0.0; V[r][q] = (r == q) ? -1.0 : 0.0; }
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) z[r][ax] = 0.0;
    }
    Seg<O> left, right;
    seg_make<O>(T[0], vw, left);
    double Pa[3], Pb[3];
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) { Pa[ax] = P[0][ax]; Pb[ax] = P[1][ax]; }
    bool spd = true;
#pragma unroll
    for (int k = 1; k < CMAX; ++k) {
        if (k < c) {
            seg_make<O>(T[k], vw, right);
            double Sm[N][N], R[N][NC];
#pragma unroll
            for (int r = 0; r < N; ++r) {
#pragma unroll
                for (int q = 0; q <= r; ++q) {
                    double v = ee_of<O>(left, r, q) + right.ss[r][q];
#pragma unroll
                    for (int j = 0; j < N; ++j) v = __builtin_fma(-left.se[j][r], W[j][q], v);
                    Sm[r][q] = v;
                }
#pragma unroll
                for (int q = 0; q < N; ++q) R[r][q] = right.se[r][q];
#pragma unroll
                for (int ax = 0; ax < 3; ++ax) {
                    double v = left.ep[r] * (Pb[ax] - Pa[ax]);
                    v = __builtin_fma(right.sp[r], P[k + 1][ax] - Pb[ax], v);
#pragma unroll
                    for (int j = 0; j < N; ++j) v = __builtin_fma(-left.se[j][r], z[j][ax], v);
                    R[r][N + ax] = v;
                }
                if (CROSS) {
#pragma unroll
                    for (int q = 0; q < N; ++q) {
                        double v = 0.0;
#pragma unroll
                        for (int j = 0; j < N; ++j) v = __builtin_fma(-left.se[j][r], V[j][q], v);
                        R[r][N + 3 + q] = v;    // V_k = -S^-1 se^T V_{k-1}
                    }
                }
            }
            spd &= SmallSpd<N, NC>::solve(Sm, R);
#pragma unroll
            for (int r = 0; r < N; ++r) {
#pragma unroll
                for (int q = 0; q < N; ++q) { W[r][q] = R[r][q]; if (CROSS) V[r][q] = R[r][N + 3 + q]; }
#pragma unroll
                for (int ax = 0; ax < 3; ++ax) z[r][ax] = R[r][N + ax];
            }
            left = right;
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) { Pa[ax] = Pb[ax]; Pb[ax] = P[k + 1][ax]; }
        }
    }
#pragma unroll
    for (int r = 0; r < N; ++r) {
#pragma unroll
        for (int q = 0; q < N; ++q) {
            double v = ee_of<O>(left, r, q), x = 0.0;
#pragma unroll
            for (int j = 0; j < N; ++j) {
                v = __builtin_fma(-left.se[j][r], W[j][q], v);
                if (CROSS) x = __builtin_fma(-left.se[j][r], V[j][q], x);
            }
            D[r][q] = v;
            if (CROSS) X[r][q] = x;
        }
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
            double v = left.ep[r] * (Pb[ax] - Pa[ax]);
#pragma unroll
            for (int j = 0; j < N; ++j) v = __builtin_fma(-left.se[j][r], z[j][ax], v);
            rr[r][ax] = v;
        }
    }
    return spd;
}

using iface::IfaceLds;
using iface::store_axis;


// The body of one wave: 64 / lpt trajectories, lpt lanes each.  `lane`'s trajectory is `bb` (caller index: boundary
// conditions, weights and status are indexed by it) of S segments starting at segment seg0 / waypoint seg0 + bb of the
// concatenated inputs; its coefficients start at element coef0 of `coeffs`.  ALIGN8: the coefficient block is only
// 8-byte aligned (fp32 storage in a mixed-order batch), so records leave in 8-byte pieces.
template <int O, typename IO, bool STATUS, bool ALIGN8>
__device__ __forceinline__ void chunked_body(const GenericArgs &a, double *lds, double *xch, int lane, int lpt, int j, bool traj_ok,
                                             int64_t bb, int64_t seg0, int S, int64_t coef0) {
    constexpr int N = O - 1, M = 2 * O;
    using IL = IfaceLds<O>;
    // lpt consecutive lanes share the trajectory (a power of two in the one-order kernel, any 1..64 in the mixed-order one);
    // j = my chunk = my position among them
    const int64_t b = bb;
    if (!traj_ok) S = 0;
    const int nch = S < lpt ? S : lpt;          // chunks in use (S >= 1 for a real trajectory)
    const int q = nch > 0 ? S / nch : 0, rem = nch > 0 ? S - q * nch : 0;
    const bool active = j < nch;
    const int c = active ? q + (j < rem ? 1 : 0) : 0;            // my segments: s0 .. s0+c-1
    const int s0 = j * q + (j < rem ? j : rem);
    const IO *wp = (const IO *)a.wp, *tm = (const IO *)a.times;
    const int64_t pt0 = seg0 + bb + s0, sg0 = seg0 + s0;
    const double vw = a.vw_per ? a.vw_per[bb] : a.vel_zero_weight;

    bool spd = true;
    double T[CMAX], P[CMAX + 1][3];
    RawChunk<IO> raw;
    // ---- step 1: the chunk's Schur complement onto its two interfaces ----
    double DR[N][N], rR[N][3];
    {
        double DL[N][N], rL[N][3], Et[N][N], unused[N][N];
        load_raw<IO>(wp, tm, pt0, sg0, c, raw);
        chunk_view<IO, true>(raw, c, T, P);
        spd &= chunk_schur<O, false>(T, P, c, vw, DL, rL, unused);       // reversed frame: the START interface row
        if (active) {
            int e = 0;
#pragma unroll
            for (int r = 0; r < N; ++r)
#pragma unroll
                for (int qq = 0; qq <= r; ++qq) lds[(e++) * 64 + lane] = ((r + qq) & 1) ? -DL[r][qq] : DL[r][qq];
#pragma unroll
            for (int r = 0; r < N; ++r)
#pragma unroll
                for (int ax = 0; ax < 3; ++ax) lds[(e++) * 64 + lane] = (r & 1) ? rL[r][ax] : -rL[r][ax];  // derivative r+1 is odd for even r
        }
        chunk_view<IO, false>(raw, c, T, P);
        spd &= chunk_schur<O, true>(T, P, c, vw, DR, rR, Et);            // end row: DR x_R + Et x_L = rR
        if (active) {
#pragma unroll
            for (int r = 0; r < N; ++r)
#pragma unroll
                for (int qq = 0; qq < N; ++qq) lds[(IL::OFF_E + r * N + qq) * 64 + lane] = Et[qq][r];       // E = Et^T
        }
    }
    __syncthreads();
    // interface i (1 <= i <= nch-1) sums chunk i-1's end row and chunk i's start row: lane i-1 adds its part
    if (active && j + 1 < nch) {
        int e = 0;
#pragma unroll
        for (int r = 0; r < N; ++r)
#pragma unroll
            for (int qq = 0; qq <= r; ++qq) { lds[e * 64 + lane + 1] += DR[r][qq]; ++e; }
#pragma unroll
        for (int r = 0; r < N; ++r)
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) { lds[e * 64 + lane + 1] += rR[r][ax]; ++e; }
    }
    __syncthreads();

    // trajectory boundary derivatives (minimum_snap.cpp:527-555): v, a given, higher ones pinned to 0
    double x0[N][3], xn[N][3];
    {
        const IO *bc = (const IO *)a.bc + (a.bc_per_traj ? bb * 12 : 0);
#pragma unroll
        for (int r = 0; r < N; ++r)
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) {
                x0[r][ax] = r == 0 ? ld(bc + 0 * 3 + ax) : r == 1 ? ld(bc + 2 * 3 + ax) : 0.0;
                xn[r][ax] = r == 0 ? ld(bc + 1 * 3 + ax) : r == 1 ? ld(bc + 3 * 3 + ax) : 0.0;
            }
    }

    // ---- step 2: twisted elimination of the interface system (minsnap_iface.h) ----
    double xL[N][3], xR[N][3];
    iface::iface_solve<O>(lds, xch, lane, j, nch, active, x0, xn, xL, xR, spd);

    // ---- step 3: the chunk as a little trajectory with every derivative known at both ends ----
    double nanacc = 0.0;
    if (active) {
        chunk_view<IO, false>(raw, c, T, P);
        double W[N][N], z[N][3];
        double Wst[CMAX][N][N], zst[CMAX][N][3];   // slot k = local waypoint k (slot 0 unused)
#pragma unroll
        for (int r = 0; r < N; ++r) {
#pragma unroll
            for (int qq = 0; qq < N; ++qq) W[r][qq] = 0.0;
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) z[r][ax] = xL[r][ax];
        }
        Seg<O> left, right;
        seg_make<O>(T[0], vw, left);
#pragma unroll
        for (int k = 1; k < CMAX; ++k) {
            if (k < c) {
                seg_make<O>(T[k], vw, right);
                double Sm[N][N], R[N][N + 3];
#pragma unroll
                for (int r = 0; r < N; ++r) {
#pragma unroll
                    for (int qq = 0; qq <= r; ++qq) {
                        double v = ee_of<O>(left, r, qq) + right.ss[r][qq];
#pragma unroll
                        for (int jx = 0; jx < N; ++jx) v = __builtin_fma(-left.se[jx][r], W[jx][qq], v);
                        Sm[r][qq] = v;
                    }
#pragma unroll
                    for (int qq = 0; qq < N; ++qq) R[r][qq] = right.se[r][qq];
#pragma unroll
                    for (int ax = 0; ax < 3; ++ax) {
                        double v = left.ep[r] * (P[k][ax] - P[k - 1][ax]);
                        v = __builtin_fma(right.sp[r], P[k + 1][ax] - P[k][ax], v);
#pragma unroll
                        for (int jx = 0; jx < N; ++jx) v = __builtin_fma(-left.se[jx][r], z[jx][ax], v);
                        R[r][N + ax] = v;
                    }
                }
                spd &= SmallSpd<N, N + 3>::solve(Sm, R);
#pragma unroll
                for (int r = 0; r < N; ++r) {
#pragma unroll
                    for (int qq = 0; qq < N; ++qq) { W[r][qq] = R[r][qq]; Wst[k][r][qq] = R[r][qq]; }
#pragma unroll
                    for (int ax = 0; ax < 3; ++ax) { z[r][ax] = R[r][N + ax]; zst[k][r][ax] = R[r][N + ax]; }
                }
                left = right;
            }
        }
        double xnx[N][3];
#pragma unroll
        for (int r = 0; r < N; ++r)
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) xnx[r][ax] = xR[r][ax];
        IO *co = (IO *)a.coeffs + coef0 + (int64_t)s0 * (3 * M);
#pragma unroll
        for (int s = CMAX - 1; s >= 0; --s) {
            if (s < c) {
                double xk[N][3];
#pragma unroll
                for (int r = 0; r < N; ++r)
#pragma unroll
                    for (int ax = 0; ax < 3; ++ax) {
                        if (s == 0) {
                            xk[r][ax] = xL[r][ax];
                        } else {
                            double v = zst[s][r][ax];
#pragma unroll
                            for (int k = 0; k < N; ++k) v = __builtin_fma(-Wst[s][r][k], xnx[k][ax], v);
                            xk[r][ax] = v;
                        }
                    }
                const double Ts = T[s];
                double ip[M], tp[N];
                ip[0] = 1.0;
                ip[1] = fast_rcp(Ts);
#pragma unroll
                for (int e = 2; e < M; ++e) ip[e] = ip[e - 1] * ip[1];
                tp[0] = Ts;
#pragma unroll
                for (int e = 1; e < N; ++e) tp[e] = tp[e - 1] * Ts;
#pragma unroll
                for (int ax = 0; ax < 3; ++ax) {
                    double xs[N], xe[N], cc[M];
#pragma unroll
                    for (int r = 0; r < N; ++r) { xs[r] = xk[r][ax]; xe[r] = xnx[r][ax]; }
                    fixedk::recover<O>(P[s][ax], P[s + 1][ax] - P[s][ax], xs, xe, tp, ip, cc);
                    if constexpr (ALIGN8 && std::is_same<IO, float>::value) {
#pragma unroll
                        for (int i = 0; i < M; i += 2)
                            *reinterpret_cast<float2 *>(co + (int64_t)s * (3 * M) + ax * M + i) = make_float2((float)cc[i], (float)cc[i + 1]);
                    } else {
                        store_axis<IO, M>(co + (int64_t)s * (3 * M) + ax * M, cc);
                    }
                    if (STATUS) {
#pragma unroll
                        for (int i = 0; i < M; ++i) nanacc = __builtin_fma((double)(IO)cc[i], 0.0, nanacc);
                    }
                }
#pragma unroll
                for (int r = 0; r < N; ++r)
#pragma unroll
                    for (int ax = 0; ax < 3; ++ax) xnx[r][ax] = xk[r][ax];
            }
        }
    }
    if (STATUS && active) {
        const int bits = (spd ? 0 : 2) | ((nanacc == 0.0) ? 0 : 1);
        if (bits) atomicOr(a.status + b, bits);
    }
}

}  // namespace chunked
}  // namespace csp
