// minsnap_twist_impl.h -- the sequential twisted sweep for the mixed-order entry (csp_minsnap_solve_mixed, BASELINE config 5):
// one LANE PAIR per trajectory of 2..64 segments, any of orders 2..5, fp64 or fp32 storage with fp64 arithmetic.
//
// Why a second family beside minsnap_chunked_impl.h: cutting a trajectory into 4-segment chunks for several lanes costs
// ~1660 VALU instructions per segment at order 4 (two Schur sweeps, an interface system every lane walks, a third sweep)
// where the plain block-LDL^T sweep of minsnap_fixed_impl.h needs ~400.  That sweep keeps every factor W_k, z_k in registers,
// which ends at 16 segments.  Here the same sweep runs in BLOCKS of K segments:
//
//   pass 1  forward elimination over all blocks but the last, keeping nothing but the running (W, z) -- written once per
//           block boundary to a per-workgroup checkpoint slot (N*N + 3N doubles per lane, coalesced; L2 resident);
//   pass 2  blocks from the middle outwards: reload the block's checkpoint, repeat its K elimination steps -- now keeping
//           W_k, z_k in registers --, back-substitute them, recover the coefficients (minimum_snap.cpp:582-591) and store.
//           The last block is only computed once (pass 1 stops before it).
//
// = two forward sweeps and one backward sweep per segment (~1.6x the plain sweep), no interface system, no spike columns.
// As in the fixed-size kernels a trajectory is split at waypoint ceil(S/2) between the two waves of a workgroup (the bottom
// role walks the time-reversed second half: reversed waypoints, odd derivatives negated) and the halves meet once through
// LDS.  The device-side bucketing (minsnap_mixed.hip) sorts by (order, S), so S is uniform inside a work unit of 64
// trajectories and every loop bound is a scalar.  Every lane stores its own records, 16 (8) bytes per instruction (no staging
// tile: at this kernel's ~1.4 TB/s of output the L2 merges the partly written lines, and a lone wave per SIMD cannot hide the
// tile's LDS round trips -- the opposite of the path kernels).
//
// Per trajectory the arithmetic is the fixed-size kernels' (same functions, same order of operations).
#pragma once
#include "minsnap_fixed_impl.h"
#include "minsnap_mixed.h"
#include "minsnap_twist_launch.h"

#include <type_traits>

#ifndef CSP_TWIST_DIRECT
#define CSP_TWIST_DIRECT 1   // 1: every lane stores its own records; 0: through a staging tile, transposed (A/B)
#endif

namespace csp {
namespace twist {

using fixedk::Seg;
using fixedk::SmallSpd;
using fixedk::ee_of;
using fixedk::seg_make;

// Output geometry: a (trajectory, segment) record of 6*O elements leaves in pieces of PB bytes (16 where the record is a whole
// number of them: blocks start 16-byte aligned, minsnap_mixed.hip block_elems; 8 for fp32 records of odd order).  The tile
// geometry (LPR lanes per record, read back transposed) serves one-segment trajectories and the CSP_TWIST_DIRECT=0 build.
template <int O, typename IO> struct Out {
    static constexpr int REC = 6 * O;
    static constexpr int RECB = REC * (int)sizeof(IO);
    static constexpr int PB = RECB % 16 == 0 ? 16 : 8;
    static constexpr int EPP = PB / (int)sizeof(IO);          // elements per piece
    static constexpr int LPR = RECB / PB;                     // lanes per record
    static constexpr int RPI = 64 / LPR;                      // records per store instruction
    static constexpr int NI = (64 + RPI - 1) / RPI;
    static constexpr int ROWB = ((LPR & 1) ? LPR : LPR + 1) * PB;   // odd number of pieces per row: conflict-free writes
    static constexpr int TILEB = 64 * ROWB;
};

template <typename IO, int PB> struct PieceT;
template <> struct PieceT<float, 16> { typedef float4 type; };
template <> struct PieceT<float, 8> { typedef float2 type; };
template <> struct PieceT<double, 16> { typedef double2 type; };

template <typename IO, int EPP> struct Pack;
template <> struct Pack<float, 4> {
    __device__ static __forceinline__ float4 make(const double *c) { return make_float4((float)c[0], (float)c[1], (float)c[2], (float)c[3]); }
};
template <> struct Pack<float, 2> {
    __device__ static __forceinline__ float2 make(const double *c) { return make_float2((float)c[0], (float)c[1]); }
};
template <> struct Pack<double, 2> {
    __device__ static __forceinline__ double2 make(const double *c) { return make_double2(c[0], c[1]); }
};

// The inputs of one block in the role's own orientation: slot i <-> local index j0 - 1 + i
// (T: segments j0-1 .. j0+K-1, P: waypoints j0-1 .. j0+K), indices clamped into the role's half.
template <int K> struct BlockIn {
    double T[K + 1];
    double P[K + 2][3];
};
// the same in the storage type, as loaded
template <int K, typename IO> struct RawIn {
    IO T[K + 1];
    IO P[K + 2][3];
};

template <int O, typename IO, bool BOTTOM, bool STATUS> struct Role {
    using G = Geo<O>;
    using OG = Out<O, IO>;
    static constexpr int N = G::N, M = G::M, K = G::K;

    const IO *wp, *tm;    // this trajectory's first waypoint / first segment time
    int S, HS;            // uniform
    double vw;
    const IO *bcp;        // boundary conditions [4][3] of this trajectory (v0, v1, a0, a1)

    // boundary velocity / acceleration of THIS role's outer end, in its orientation -- read where they are needed (the first
    // block of each pass and the very last step) instead of held: twelve registers for the life of the unit otherwise
    __device__ __forceinline__ double bc_at(int r, int ax) const {
        return r == 0 ? (BOTTOM ? -(double)bcp[1 * 3 + ax] : (double)bcp[0 * 3 + ax]) : r == 1 ? (BOTTOM ? (double)bcp[3 * 3 + ax] : (double)bcp[2 * 3 + ax]) : 0.0;
    }

    __device__ __forceinline__ void load_raw(int j0, RawIn<K, IO> &raw) const {
#pragma unroll
        for (int i = 0; i <= K; ++i) {
            int jj = j0 - 1 + i;
            jj = jj < 0 ? 0 : (jj > HS - 1 ? HS - 1 : jj);
            raw.T[i] = tm[BOTTOM ? S - 1 - jj : jj];
        }
#pragma unroll
        for (int i = 0; i <= K + 1; ++i) {
            int jj = j0 - 1 + i;
            jj = jj < 0 ? 0 : (jj > HS ? HS : jj);
            const IO *p = wp + (BOTTOM ? S - jj : jj) * 3;
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) raw.P[i][ax] = p[ax];
        }
    }
    __device__ __forceinline__ void load_in(int j0, BlockIn<K> &in) const {
        RawIn<K, IO> raw;
        load_raw(j0, raw);
        widen(raw, in);
    }
    __device__ static __forceinline__ void widen(const RawIn<K, IO> &raw, BlockIn<K> &in) {
#pragma unroll
        for (int i = 0; i <= K; ++i) in.T[i] = (double)raw.T[i];
#pragma unroll
        for (int i = 0; i <= K + 1; ++i)
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) in.P[i][ax] = (double)raw.P[i][ax];
    }

    // ip[e] = T^-e: all that is carried from one step to the next of a segment (a whole Seg is 2N^2 + 2N doubles; rebuilding it
    // from the ladder costs ~30 multiplications a step and frees a tenth of the register file at order 5)
    __device__ static __forceinline__ void ladder(double T, double (&ip)[M]) {
        ip[0] = 1.0;
        ip[1] = fast_rcp(T);
#pragma unroll
        for (int e = 2; e < M; ++e) ip[e] = ip[e - 1] * ip[1];
    }
    __device__ __forceinline__ void seg_of(const double (&ip)[M], Seg<O> &s) const {   // = fixedk::seg_make after its ladder
#pragma unroll
        for (int r = 0; r < N; ++r) {
#pragma unroll
            for (int c = 0; c < N; ++c) {
                s.ss[r][c] = Tab<O>::QT(r + 1, c + 1) * ip[M - 3 - r - c];
                s.se[r][c] = Tab<O>::QT(r + 1, O + c + 1) * ip[M - 3 - r - c];
            }
            s.sp[r] = Tab<O>::QT(r + 1, 0) * ip[M - 2 - r];
            s.ep[r] = Tab<O>::QT(O + r + 1, 0) * ip[M - 2 - r];
        }
        s.ss[0][0] += vw;
    }

    // K elimination steps from local waypoint j0 on (minimum_snap.cpp:564-566, as minsnap_fixed_impl.h).  On entry W, z are
    // the factors of waypoint j0 - 1 (j0 = 0: W = 0, z = the boundary derivatives) and ipL the ladder of segment j0 - 1
    // (RELEFT: rebuilt here from its time).  The test against HS is a real branch on purpose: it ends the scheduling region
    // once per segment (minsnap_fixed_path_impl.h has the measurements).
    template <bool STORE, bool RELEFT>
    __device__ __forceinline__ void forward(int j0, const BlockIn<K> &in, double (&W)[N][N], double (&z)[N][3],
                                            double (&dPl)[3], bool &spd, double (&Wst)[K][N][N], double (&zst)[K][N][3]) const {
#pragma unroll
        for (int kk = 0; kk < K; ++kk) {
            const int j = j0 + kk;
            if (j < HS) {
                double ipR[M];
                ladder(in.T[kk + 1], ipR);
                if (kk > 0 || j0 >= 1) {
                    // the left segment whole, of the right one only what enters here (ss, sp; its se block goes straight into R
                    // afterwards): two whole Segs at once are 160 registers at order 5
                    double Sm[N][N], R[N][N + 3];   // right-hand sides: [C_j | y_j]
                    {
                        // the left segment's ladder is rebuilt from its time (one reciprocal, M - 2 products) rather than
                        // carried over from the step before: M doubles less across the solve
                        double ipL[M];
                        ladder(in.T[kk], ipL);
                        Seg<O> left;
                        seg_of(ipL, left);
#pragma unroll
                        for (int r = 0; r < N; ++r) {
#pragma unroll
                            for (int c = 0; c <= r; ++c) {
                                double rss = Tab<O>::QT(r + 1, c + 1) * ipR[M - 3 - r - c];
                                if (r == 0 && c == 0) rss += vw;
                                double v = ee_of<O>(left, r, c) + rss;
#pragma unroll
                                for (int q = 0; q < N; ++q) v = __builtin_fma(-left.se[q][r], W[q][c], v);
                                Sm[r][c] = v;
                            }
                            const double rsp = Tab<O>::QT(r + 1, 0) * ipR[M - 2 - r];
#pragma unroll
                            for (int ax = 0; ax < 3; ++ax) {
                                double v = left.ep[r] * (in.P[kk + 1][ax] - in.P[kk][ax]);
                                v = __builtin_fma(rsp, in.P[kk + 2][ax] - in.P[kk + 1][ax], v);
#pragma unroll
                                for (int q = 0; q < N; ++q) v = __builtin_fma(-left.se[q][r], z[q][ax], v);
                                R[r][N + ax] = v;
                            }
                        }
                    }
#pragma unroll
                    for (int r = 0; r < N; ++r)
#pragma unroll
                        for (int c = 0; c < N; ++c) R[r][c] = Tab<O>::QT(r + 1, O + c + 1) * ipR[M - 3 - r - c];
                    spd &= SmallSpd<N, N + 3>::solve(Sm, R);
#pragma unroll
                    for (int r = 0; r < N; ++r) {
#pragma unroll
                        for (int c = 0; c < N; ++c) { W[r][c] = R[r][c]; if (STORE) Wst[kk][r][c] = R[r][c]; }
#pragma unroll
                        for (int ax = 0; ax < 3; ++ax) { z[r][ax] = R[r][N + ax]; if (STORE) zst[kk][r][ax] = R[r][N + ax]; }
                    }
                }
#pragma unroll
                for (int ax = 0; ax < 3; ++ax) dPl[ax] = in.P[kk + 2][ax] - in.P[kk + 1][ax];
            }
        }
    }
};

// One role of one work unit.  tile: this wave's staging tile; x_mine / x_other: the exchange areas (this role writes the
// first, reads the second); s_coef[row]: element offset of row's coefficient block; ck: this (workgroup, role)'s checkpoints.
template <int O, typename IO, bool BOTTOM, bool STATUS>
__device__ __forceinline__ void twist_role(const GenericArgs &a, int lane, int S, int rows, int64_t bb, int64_t seg0, char *tile,
                                           double *x_mine, const double *x_other, const long long *s_coef, double *ck, long long coef0) {
    using RT = Role<O, IO, BOTTOM, STATUS>;
    using G = Geo<O>;
    using OG = Out<O, IO>;
    constexpr int N = G::N, M = G::M, K = G::K;
    RT ro;
    ro.S = S;
    ro.HS = BOTTOM ? S / 2 : (S + 1) / 2;
    ro.wp = (const IO *)a.wp + (seg0 + bb) * 3;
    ro.tm = (const IO *)a.times + seg0;
    ro.vw = a.vw_per ? a.vw_per[bb] : a.vel_zero_weight;
    ro.bcp = (const IO *)a.bc + (a.bc_per_traj ? bb * 12 : 0);
    const int HS = ro.HS, nb = (HS + K - 1) / K;

    double W[N][N], z[N][3], dPl[3] = {0.0, 0.0, 0.0};
    double Wst[K][N][N], zst[K][N][3];
    const IO t_last = ro.tm[BOTTOM ? S - ro.HS : ro.HS - 1];   // the segment next to the middle waypoint: its ladder is needed once
    bool spd = true;
    BlockIn<K> in;
#pragma unroll
    for (int r = 0; r < N; ++r) {
#pragma unroll
        for (int c = 0; c < N; ++c) W[r][c] = 0.0;
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) z[r][ax] = ro.bc_at(r, ax);
    }
    // ---- pass 1: forward over blocks 0 .. nb-2, a checkpoint at every block boundary ----
    for (int blk = 0; blk < nb - 1; ++blk) {
        ro.load_in(blk * K, in);
        ro.template forward<false, false>(blk * K, in, W, z, dPl, spd, Wst, zst);
        double *c = ck + (size_t)blk * G::CKD * 64 + lane;
        int e = 0;
#pragma unroll
        for (int r = 0; r < N; ++r)
#pragma unroll
            for (int q = 0; q < N; ++q) c[(e++) * 64] = W[r][q];
#pragma unroll
        for (int r = 0; r < N; ++r)
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) c[(e++) * 64] = z[r][ax];
    }
    // ---- pass 2: blocks nb-1 .. 0: repeat the block's steps keeping the factors, back-substitute, recover, store ----
    double xn[N][3];
    double nanacc = 0.0;
    const int grp = lane / OG::LPR, lin = lane - grp * OG::LPR;
    typedef typename PieceT<IO, OG::PB>::type piece_t;
    constexpr bool DIRECT = CSP_TWIST_DIRECT != 0;
    IO *co_own = reinterpret_cast<IO *>(a.coeffs) + coef0;
    for (int blk = nb - 1; blk >= 0; --blk) {
        const int j0 = blk * K;
        ro.load_in(j0, in);
        if (blk < nb - 1) {   // (the last block continues from pass 1's registers)
            if (blk >= 1) {
                const double *c = ck + (size_t)(blk - 1) * G::CKD * 64 + lane;
                int e = 0;
#pragma unroll
                for (int r = 0; r < N; ++r)
#pragma unroll
                    for (int q = 0; q < N; ++q) W[r][q] = c[(e++) * 64];
#pragma unroll
                for (int r = 0; r < N; ++r)
#pragma unroll
                    for (int ax = 0; ax < 3; ++ax) z[r][ax] = c[(e++) * 64];
            } else {
#pragma unroll
                for (int r = 0; r < N; ++r) {
#pragma unroll
                    for (int c = 0; c < N; ++c) W[r][c] = 0.0;
#pragma unroll
                    for (int ax = 0; ax < 3; ++ax) z[r][ax] = ro.bc_at(r, ax);
                }
            }
        }
        ro.template forward<true, true>(j0, in, W, z, dPl, spd, Wst, zst);
        if (blk == nb - 1) {
            // Schur carry of this half onto the middle waypoint, exchanged through LDS (minsnap_fixed_impl.h)
            double Cm[N][N], cm[N][3];
            Seg<O> left;
            double ipL[M];
            RT::ladder((double)t_last, ipL);
            ro.seg_of(ipL, left);
#pragma unroll
            for (int r = 0; r < N; ++r) {
#pragma unroll
                for (int c = 0; c <= r; ++c) {
                    double v = ee_of<O>(left, r, c);
#pragma unroll
                    for (int q = 0; q < N; ++q) v = __builtin_fma(-left.se[q][r], W[q][c], v);
                    Cm[r][c] = v;
                }
#pragma unroll
                for (int ax = 0; ax < 3; ++ax) {
                    double v = left.ep[r] * dPl[ax];
#pragma unroll
                    for (int q = 0; q < N; ++q) v = __builtin_fma(-left.se[q][r], z[q][ax], v);
                    cm[r][ax] = v;
                }
            }
            {
                int e = 0;
#pragma unroll
                for (int r = 0; r < N; ++r)
#pragma unroll
                    for (int c = 0; c <= r; ++c) x_mine[(e++) * 64 + lane] = Cm[r][c];
#pragma unroll
                for (int r = 0; r < N; ++r)
#pragma unroll
                    for (int ax = 0; ax < 3; ++ax) x_mine[(e++) * 64 + lane] = cm[r][ax];
            }
            fixedk::lds_barrier();
            {
                double Sm[N][N], R[N][3];
                int e = 0;
#pragma unroll
                for (int r = 0; r < N; ++r)
#pragma unroll
                    for (int c = 0; c <= r; ++c) {
                        const double o = x_other[(e++) * 64 + lane];
                        Sm[r][c] = Cm[r][c] + (((r + c) & 1) ? -o : o);   // the other side's carry, conjugated by the odd-derivative sign flip
                    }
#pragma unroll
                for (int r = 0; r < N; ++r)
#pragma unroll
                    for (int ax = 0; ax < 3; ++ax) {
                        const double o = x_other[(e++) * 64 + lane];
                        R[r][ax] = cm[r][ax] + ((r & 1) ? o : -o);
                    }
                spd &= SmallSpd<N, 3>::solve(Sm, R);
#pragma unroll
                for (int r = 0; r < N; ++r)
#pragma unroll
                    for (int ax = 0; ax < 3; ++ax) xn[r][ax] = R[r][ax];
            }
        }
        // back-substitution fused with the Hermite -> monomial recovery, local segments of this block downwards
#pragma unroll
        for (int kk = K - 1; kk >= 0; --kk) {
            const int j = j0 + kk;
            if (j < HS) {
                double xk[N][3];
                if (kk == 0 && j0 == 0) {
#pragma unroll
                    for (int r = 0; r < N; ++r)
#pragma unroll
                        for (int ax = 0; ax < 3; ++ax) xk[r][ax] = ro.bc_at(r, ax);
                } else {
#pragma unroll
                    for (int r = 0; r < N; ++r)
#pragma unroll
                        for (int ax = 0; ax < 3; ++ax) {
                            double v = zst[kk][r][ax];
#pragma unroll
                            for (int c = 0; c < N; ++c) v = __builtin_fma(-Wst[kk][r][c], xn[c][ax], v);
                            xk[r][ax] = v;
                        }
                }
                const double Tj = in.T[kk + 1];
                double ip[M], tp[N];
                ip[0] = 1.0;
                ip[1] = fast_rcp(Tj);
#pragma unroll
                for (int e = 2; e < M; ++e) ip[e] = ip[e - 1] * ip[1];
                tp[0] = Tj;
#pragma unroll
                for (int e = 1; e < N; ++e) tp[e] = tp[e - 1] * Tj;
                const int g = BOTTOM ? S - 1 - j : j;   // global segment
#pragma unroll
                for (int ax = 0; ax < 3; ++ax) {
                    double xs[N], xe[N], cc[M];
#pragma unroll
                    for (int r = 0; r < N; ++r) {
                        const double sgn = (BOTTOM && !(r & 1)) ? -1.0 : 1.0;
                        xs[r] = BOTTOM ? sgn * xn[r][ax] : xk[r][ax];
                        xe[r] = BOTTOM ? sgn * xk[r][ax] : xn[r][ax];
                    }
                    const double Plo = in.P[kk + 1][ax], Phi = in.P[kk + 2][ax];
                    const double Ps = BOTTOM ? Phi : Plo, Pe = BOTTOM ? Plo : Phi;
                    fixedk::recover<O>(Ps, Pe - Ps, xs, xe, tp, ip, cc);
                    if (DIRECT) {
                        // the lane stores its own record: 64 partly written lines per instruction, completed by the next
                        // instructions of the same lane (the L2 merges them) -- no tile, no transposed read-back, and no
                        // predicate: the lanes beyond a ragged unit shadow row 0 (same inputs, same arithmetic), so they
                        // store row 0's bytes a second time
                        IO *dst = co_own + (long long)g * OG::REC + ax * M;
#pragma unroll
                        for (int i = 0; i < M; i += OG::EPP) *reinterpret_cast<piece_t *>(dst + i) = Pack<IO, OG::EPP>::make(cc + i);
                    } else {
#pragma unroll
                        for (int i = 0; i < M; i += OG::EPP)
                            *reinterpret_cast<piece_t *>(tile + lane * OG::ROWB + (ax * M + i) * (int)sizeof(IO)) = Pack<IO, OG::EPP>::make(cc + i);
                    }
                    if (STATUS) {
#pragma unroll
                        for (int i = 0; i < M; ++i) nanacc = __builtin_fma((double)(IO)cc[i], 0.0, nanacc);
                    }
                }
                if (!DIRECT) {
                // LDS operations of one wave execute in order: no barrier, only compiler fences around the transposed read
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                for (int it = 0; it < OG::NI; ++it) {
                    const int row = it * OG::RPI + grp;
                    if (lane < OG::RPI * OG::LPR && row < rows) {
                        const piece_t v = *reinterpret_cast<const piece_t *>(tile + row * OG::ROWB + lin * OG::PB);
                        char *dst = reinterpret_cast<char *>(a.coeffs) + (s_coef[row] + (long long)g * OG::REC) * (long long)sizeof(IO) + lin * OG::PB;
                        *reinterpret_cast<piece_t *>(dst) = v;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                }
#pragma unroll
                for (int r = 0; r < N; ++r)
#pragma unroll
                    for (int ax = 0; ax < 3; ++ax) xn[r][ax] = xk[r][ax];
            }
        }
    }
    if (STATUS && lane < rows) {
        const int bits = (spd ? 0 : 2) | ((nanacc == 0.0) ? 0 : 1);
        if (bits) atomicOr(a.status + bb, bits);
    }
}

// A one-segment trajectory has no free derivative at all: both ends are boundary conditions.  Top wave only.
template <int O, typename IO, bool STATUS>
__device__ __forceinline__ void single_segment(const GenericArgs &a, int lane, int rows, int64_t bb, int64_t seg0, char *tile,
                                               const long long *s_coef) {
    using OG = Out<O, IO>;
    constexpr int N = O - 1, M = 2 * O;
    typedef typename PieceT<IO, OG::PB>::type piece_t;
    const IO *wp = (const IO *)a.wp + (seg0 + bb) * 3, *bc = (const IO *)a.bc + (a.bc_per_traj ? bb * 12 : 0);
    const double Tj = (double)((const IO *)a.times)[seg0];
    double ip[M], tp[N];
    ip[0] = 1.0;
    ip[1] = fast_rcp(Tj);
#pragma unroll
    for (int e = 2; e < M; ++e) ip[e] = ip[e - 1] * ip[1];
    tp[0] = Tj;
#pragma unroll
    for (int e = 1; e < N; ++e) tp[e] = tp[e - 1] * Tj;
    double nanacc = 0.0;
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
        double xs[N], xe[N], cc[M];
#pragma unroll
        for (int r = 0; r < N; ++r) {
            xs[r] = r == 0 ? (double)bc[0 * 3 + ax] : r == 1 ? (double)bc[2 * 3 + ax] : 0.0;
            xe[r] = r == 0 ? (double)bc[1 * 3 + ax] : r == 1 ? (double)bc[3 * 3 + ax] : 0.0;
        }
        const double Ps = (double)wp[ax], Pe = (double)wp[3 + ax];
        fixedk::recover<O>(Ps, Pe - Ps, xs, xe, tp, ip, cc);
#pragma unroll
        for (int i = 0; i < M; i += OG::EPP)
            *reinterpret_cast<piece_t *>(tile + lane * OG::ROWB + (ax * M + i) * (int)sizeof(IO)) = Pack<IO, OG::EPP>::make(cc + i);
        if (STATUS) {
#pragma unroll
            for (int i = 0; i < M; ++i) nanacc = __builtin_fma((double)(IO)cc[i], 0.0, nanacc);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int grp = lane / OG::LPR, lin = lane - grp * OG::LPR;
#pragma unroll
    for (int it = 0; it < OG::NI; ++it) {
        const int row = it * OG::RPI + grp;
        if (lane < OG::RPI * OG::LPR && row < rows) {
            const piece_t v = *reinterpret_cast<const piece_t *>(tile + row * OG::ROWB + lin * OG::PB);
            char *dst = reinterpret_cast<char *>(a.coeffs) + s_coef[row] * (long long)sizeof(IO) + lin * OG::PB;
            *reinterpret_cast<piece_t *>(dst) = v;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    if (STATUS && lane < rows && nanacc != 0.0) atomicOr(a.status + bb, 1);
}

static __constant__ TwistCostOrder g_cost_order = make_twist_cost_order();

// LDS of one workgroup, carved per order: [tile top | tile bottom | exchange top | exchange bottom]
template <int O, typename IO> struct UnitLds {
    static constexpr int TILEB = Out<O, IO>::TILEB;
    static constexpr int XCHB = Geo<O>::CARRY * 64 * 8;
    static constexpr int BYTES = 2 * TILEB + 2 * XCHB;
};
template <typename IO> constexpr int unit_lds_bytes() {
    int m = UnitLds<2, IO>::BYTES;
    if (UnitLds<3, IO>::BYTES > m) m = UnitLds<3, IO>::BYTES;
    if (UnitLds<4, IO>::BYTES > m) m = UnitLds<4, IO>::BYTES;
    if (UnitLds<5, IO>::BYTES > m) m = UnitLds<5, IO>::BYTES;
    return m;
}

// One work unit: 64 trajectories (`rows` of them live) of order O and S segments, `first` = their place in perm
template <int O, typename IO, bool STATUS>
__device__ __forceinline__ void run_unit(const GenericArgs &a, const int32_t *perm, const int64_t *coef_off, char *lds, long long *s_coef,
                                         double *ck, int wave, int lane, int S, int first, int rows) {
    using UL = UnitLds<O, IO>;
    char *tile0 = lds, *tile1 = lds + UL::TILEB;
    double *x0 = reinterpret_cast<double *>(lds + 2 * UL::TILEB), *x1 = reinterpret_cast<double *>(lds + 2 * UL::TILEB + UL::XCHB);
    const int64_t bb = perm[first + (lane < rows ? lane : 0)];   // idle lanes shadow row 0 (loads only)
    const int64_t seg0 = a.seg_off[bb];
    const long long coef0 = coef_off[bb];
    if (wave == 0) s_coef[lane] = coef0;
    if (S == 1) {
        if (wave == 0) single_segment<O, IO, STATUS>(a, lane, rows, bb, seg0, tile0, s_coef);
    } else if (wave == 0) {
        twist_role<O, IO, false, STATUS>(a, lane, S, rows, bb, seg0, tile0, x0, x1, s_coef, ck, coef0);
    } else {
        twist_role<O, IO, true, STATUS>(a, lane, S, rows, bb, seg0, tile1, x1, x0, s_coef, ck, coef0);
    }
}

// ONE persistent launch for all orders: the (order, S) classes are listed by descending unit cost (TwistCostOrder); workgroup
// w starts with unit w and then pulls the next unstarted one from a counter -- list scheduling: the heaviest units (a
// 64-segment unit of order 5 alone is as long as an average workgroup's whole share at B = 65536) start first and nothing
// queues behind them.  One atomic per unit and workgroup, spread over the run (the counter that cost 1.2 ms in the first
// version of the chunked launch was hit by 20 k waves at once).  Exit: every workgroup leaves when the counter passes the
// total.
template <typename IO, bool STATUS>
__global__ void __launch_bounds__(128) minsnap_twist_kernel(GenericArgs a, const int32_t *perm, const int64_t *coef_off, MixedTable *tab,
                                                            double *ckws, size_t ck_role_doubles) {
    constexpr int NK = 4 * MIXED_NCLS;
    __shared__ __attribute__((aligned(16))) char lds[unit_lds_bytes<IO>()];
    __shared__ long long s_coef[64];
    __shared__ int s_ustart[NK + 1];
    __shared__ int s_bstart[4][MIXED_NCLS + 1];
    __shared__ int s_u;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int q = tid; q <= NK; q += 128) s_ustart[q] = tab->tw_ustart[q];
    for (int q = tid; q < 4 * (MIXED_NCLS + 1); q += 128) s_bstart[q / (MIXED_NCLS + 1)][q % (MIXED_NCLS + 1)] = tab->bucket_start[q / (MIXED_NCLS + 1)][q % (MIXED_NCLS + 1)];
    __syncthreads();
    const int total = s_ustart[NK];
    double *ck = ckws + ((size_t)blockIdx.x * 2 + wave) * ck_role_doubles;
    int u = blockIdx.x;
    while (u < total) {
        int p = 0;   // the class of unit u: the last position with tw_ustart[p] <= u
#pragma unroll
        for (int step = NK / 2; step >= 1; step >>= 1)
            if (s_ustart[p + step] <= u) p += step;
        p = __builtin_amdgcn_readfirstlane(p);
        const int key = g_cost_order.key_at[p];
        const int oi = key / MIXED_NCLS, k = key % MIXED_NCLS;
        const int S = SMAX - k;
        const int first = __builtin_amdgcn_readfirstlane(s_bstart[oi][k] + (u - s_ustart[p]) * 64);
        const int left_in_bucket = __builtin_amdgcn_readfirstlane(s_bstart[oi][k + 1]) - first;
        const int rows = left_in_bucket < 64 ? left_in_bucket : 64;
        switch (oi) {
            case 0: run_unit<2, IO, STATUS>(a, perm, coef_off, lds, s_coef, ck, wave, lane, S, first, rows); break;
            case 1: run_unit<3, IO, STATUS>(a, perm, coef_off, lds, s_coef, ck, wave, lane, S, first, rows); break;
            case 2: run_unit<4, IO, STATUS>(a, perm, coef_off, lds, s_coef, ck, wave, lane, S, first, rows); break;
            default: run_unit<5, IO, STATUS>(a, perm, coef_off, lds, s_coef, ck, wave, lane, S, first, rows); break;
        }
        fixedk::lds_barrier();   // tiles, exchange areas and s_coef are reused by the next unit
        // (claiming one unit AHEAD, to hide the counter's round trip, was measured: every workgroup then holds a second unit
        // from t = 0 on, the one with the heaviest first unit included -- 179 -> 217 us)
        if (tid == 0) s_u = atomicAdd(&tab->next_unit, 1) + (int)gridDim.x;
        __syncthreads();
        u = s_u;
    }
}

}  // namespace twist
}  // namespace csp

// one translation unit per (storage type, status) so that the four variants compile in parallel
#define CSP_TWIST_INSTANTIATE(IO, STATUS, NAME)                                                                                          \
    namespace csp {                                                                                                                      \
    namespace twist {                                                                                                                    \
    hipError_t NAME(const GenericArgs &a, const int32_t *perm, const int64_t *coef_off, MixedTable *tab, double *ckws,                   \
                    size_t ck_role_doubles, int workgroups, hipStream_t st) {                                                            \
        hipLaunchKernelGGL((minsnap_twist_kernel<IO, STATUS>), dim3((unsigned)workgroups), dim3(128), 0, st, a, perm, coef_off, tab,     \
                           ckws, ck_role_doubles);                                                                                       \
        return hipGetLastError();                                                                                                        \
    }                                                                                                                                    \
    }                                                                                                                                    \
    }
