// minsnap_fixed_impl.h -- register-resident kernels for the uniform fixed-size buckets
// (fp64, derivative order O in 2..5, 2 <= S <= 16 segments of either parity, no path penalty).
// Instantiated once per order by minsnap_fixed_o<O>.hip; the dispatcher is minsnap_fixed.hip.
//
// Mapping: a trajectory is split at its middle waypoint between two WAVES of one
// workgroup: the top role owns the first ceil(S/2) segments, the bottom role the other floor(S/2).
// Wave 0 ("top") eliminates its interior waypoints downwards, wave 1
// ("bottom") runs the very same code on the time-reversed second half (reversed waypoint order,
// odd derivatives negated), i.e. a twisted block-LDL^T factorisation of the block-tridiagonal
// R_PP (minimum_snap.cpp:564-566).  Lane l of both waves owns trajectory 64*slice+l, so every
// value a lane needs later (W_k = S_k^-1 C_k and z_k = S_k^-1 y_k) stays in ITS registers --
// nothing is spilled to memory between the forward and the backward sweep.  The two halves meet
// once, through n(n+1)/2 + 3n doubles per lane in LDS (n = O-1): each side's Schur carry onto the
// middle waypoint.  The roles are wave-uniform, so the only divergence is a scalar branch.
//
// Algorithmic HBM traffic per trajectory: 8*(3(S+1)+S) bytes in, 8*6*O*S bytes out
// (O=4, S=16: 536 + 3072 = 3608 B); no workspace.
#pragma once
#include "minsnap_device.h"
#include "minsnap_launch.h"

#include <cstdlib>
#include <type_traits>

#ifdef CSP_STAMPS
// Diagnostic build only (python cs-pathplan_amd/build.py --stamps): per-wave s_memtime stamps
// written to a buffer nothing else reads.  The shipped library contains none of this.
static __device__ unsigned long long csp_g_stamps[8192 * 8];  // one copy per translation unit; tools read order 4's
#define CSP_STAMP(slot)                                                                     \
    do {                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                  \
        unsigned long long t_ = __builtin_amdgcn_s_memtime();                               \
        __builtin_amdgcn_sched_barrier(0);                                                  \
        if ((threadIdx.x & 63) == 0 && blockIdx.x < 4096)                                   \
            csp_g_stamps[(blockIdx.x * 2 + (threadIdx.x >> 6)) * 8 + (slot)] = t_;          \
    } while (0)
#define CSP_STAMP_RT(slot)                                                                  \
    do {                                                                                    \
        unsigned long long t_ = __builtin_amdgcn_s_memrealtime();                           \
        if ((threadIdx.x & 63) == 0 && blockIdx.x < 4096)                                   \
            csp_g_stamps[(blockIdx.x * 2 + (threadIdx.x >> 6)) * 8 + (slot)] = t_;          \
    } while (0)
#else
#define CSP_STAMP(slot) do { } while (0)
#define CSP_STAMP_RT(slot) do { } while (0)
#endif

namespace csp {
namespace fixedk {

// Workgroup barrier that orders LDS traffic only.  __syncthreads() would also wait for every
// outstanding global store (vmcnt(0)); the persistent kernel keeps stores and the next
// slice's LDS-DMA in flight across its barriers.
// 16-byte coefficient store.  `nt` (wave-uniform, GenericArgs::nt_stores): non-temporal, for batches whose coefficients
// exceed the Infinity Cache -- B = 524288: 348 us against 378 us (67.9 % against 62.5 % of HBM peak); at B = 65536, whose
// 201 MB the cache absorbs, the ordinary store is the faster one (42.5 us against 46.1 us), so the launcher decides.
typedef double v2d_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void store16(char *p, const double2 &v, bool nt) {
    if (nt) {
        v2d_t x = {v.x, v.y};
        __builtin_nontemporal_store(x, reinterpret_cast<v2d_t *>(p));
    } else {
        *reinterpret_cast<double2 *>(p) = v;
    }
}

__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Scaled per-segment constants (free derivatives r = 1..O-1 -> index r-1).
// ee[r][c] = (-1)^(r+c) ss[r][c] and Qt[.][end pos] = -Qt[.][start pos] (checked in
// tests/test_tables.py), so only ss, se and the two start-position columns are formed.
template <int O> struct Seg {
    static constexpr int N = O - 1;
    double ss[N][N];  // symmetric; full storage keeps the unrolled code simple
    double se[N][N];
    double sp[N];     // Qt[start r][start pos]
    double ep[N];     // Qt[end r][start pos]
};

template <int O> __device__ __forceinline__ void seg_make(double T, double vw, Seg<O> &s) {
    constexpr int N = O - 1, M = 2 * O;
    double ip[M];  // ip[e] = T^-e;  Qt(T)[a][b] = Qt1[a][b] * T^(1 - 2o + d_a + d_b)
    ip[0] = 1.0;
    ip[1] = fast_rcp(T);
#pragma unroll
    for (int e = 2; e < M; ++e) ip[e] = ip[e - 1] * ip[1];
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
    s.ss[0][0] += vw;  // zero-velocity penalty: +w on the velocity diagonal (minimum_snap.cpp:473-509)
}

template <int O> __device__ __forceinline__ double ee_of(const Seg<O> &s, int r, int c) {
    return ((r + c) & 1) ? -s.ss[r][c] : s.ss[r][c];
}

// X = S^-1 B for a small symmetric positive definite S (lower triangle read), NR right-hand
// sides.  n <= 3: cofactor inverse with ONE Newton-refined reciprocal (short dependency chain);
// n = 4: LDL^T.  Returns false when a leading minor / pivot is not positive.
template <int N, int NR> struct SmallSpd;
template <int NR> struct SmallSpd<1, NR> {
    __device__ static __forceinline__ bool solve(double (&S)[1][1], double (&B)[1][NR]) {
        const double r = fast_rcp(S[0][0]);
#pragma unroll
        for (int c = 0; c < NR; ++c) B[0][c] *= r;
        return S[0][0] > 0.0;
    }
};
template <int NR> struct SmallSpd<2, NR> {
    __device__ static __forceinline__ bool solve(double (&S)[2][2], double (&B)[2][NR]) {
        const double a = S[0][0], b = S[1][0], c = S[1][1];
        const double det = __builtin_fma(a, c, -b * b);
        const double rd = fast_rcp(det);
        const double i00 = c * rd, i10 = -b * rd, i11 = a * rd;
#pragma unroll
        for (int k = 0; k < NR; ++k) {
            const double x0 = B[0][k], x1 = B[1][k];
            B[0][k] = __builtin_fma(i10, x1, i00 * x0);
            B[1][k] = __builtin_fma(i11, x1, i10 * x0);
        }
        return (a > 0.0) && (det > 0.0);
    }
};
template <int NR> struct SmallSpd<3, NR> {
    __device__ static __forceinline__ bool solve(double (&S)[3][3], double (&B)[3][NR]) {
        const double a = S[0][0], b = S[1][0], c = S[1][1], d = S[2][0], e = S[2][1], f = S[2][2];
        const double c00 = __builtin_fma(c, f, -e * e);
        const double c10 = __builtin_fma(d, e, -b * f);
        const double c20 = __builtin_fma(b, e, -c * d);
        const double c11 = __builtin_fma(a, f, -d * d);
        const double c21 = __builtin_fma(b, d, -a * e);
        const double c22 = __builtin_fma(a, c, -b * b);
        const double det = __builtin_fma(a, c00, __builtin_fma(b, c10, d * c20));
        const double rd = fast_rcp(det);
        const double i00 = c00 * rd, i10 = c10 * rd, i20 = c20 * rd, i11 = c11 * rd, i21 = c21 * rd, i22 = c22 * rd;
#pragma unroll
        for (int k = 0; k < NR; ++k) {
            const double x0 = B[0][k], x1 = B[1][k], x2 = B[2][k];
            B[0][k] = __builtin_fma(i20, x2, __builtin_fma(i10, x1, i00 * x0));
            B[1][k] = __builtin_fma(i21, x2, __builtin_fma(i11, x1, i10 * x0));
            B[2][k] = __builtin_fma(i22, x2, __builtin_fma(i21, x1, i20 * x0));
        }
        return (a > 0.0) && (c22 > 0.0) && (det > 0.0);
    }
};
template <int NR> struct SmallSpd<4, NR> {
    __device__ static __forceinline__ bool solve(double (&S)[4][4], double (&B)[4][NR]) {
        return spd_solve<4, NR, double>(S, B) > 0.0;
    }
};

// Hermite -> monomial map of one segment and axis.  xs/xe: free derivatives at the segment's
// start/end in GLOBAL orientation, dP = P_end - P_start, tp[r] = T^(r+1), ip[e] = T^-e.
template <int O>
__device__ __forceinline__ void recover(double Ps, double dP, const double (&xs)[O - 1], const double (&xe)[O - 1],
                                        const double (&tp)[O - 1], const double (&ip)[2 * O], double (&c)[2 * O]) {
    constexpr int N = O - 1, M = 2 * O;
    double hs[N], he[N];
#pragma unroll
    for (int r = 0; r < N; ++r) { hs[r] = xs[r] * tp[r]; he[r] = xe[r] * tp[r]; }
#pragma unroll
    for (int i = 0; i < O; ++i) {
        // G[i][0] + G[i][O] = 0 for the high rows: positions enter through dP only
        double acc = Tab<O>::G(i, O) * dP;
#pragma unroll
        for (int r = 0; r < N; ++r) {
            acc = __builtin_fma(Tab<O>::G(i, r + 1), hs[r], acc);
            acc = __builtin_fma(Tab<O>::G(i, O + r + 1), he[r], acc);
        }
        c[i] = acc * ip[M - 1 - i];
    }
#pragma unroll
    for (int j = 1; j < O; ++j) c[M - 1 - j] = xs[j - 1] * Tab<O>::G(M - 1 - j, j);  // derivative j / j!
    c[M - 1] = Ps;
}


// ---- whole-line coefficient stores (round 3) -----------------------------------------------------------------------
// A role's records of one trajectory are one contiguous byte range [LO, HI) of the coeffs array, produced record by
// record (top role: descending segments, bottom role: ascending).  Records of 96 / 144 / 240 bytes (orders 2 / 3 / 5) do not
// end on 128-byte lines, so a record-at-a-time store leaves lines shared between two store instructions issued a whole
// segment apart: the L2 has to merge the halves (WRITE_SIZE 1.08-1.14x the coefficients) and non-temporal stores, which
// are not merged, lose 1.3-2x.  Here each staging-tile row is a RING of trajectory bytes (position = byte offset mod RINGB):
// a record is written into the ring, every 128-byte line it COMPLETES is stored whole (8 rows x 128 bytes per store
// instruction, 8 lanes per line), and the bytes of a line still waiting for the next record stay in the ring (< 128 of
// them, so RINGB = record + held bytes rounded up to a line).  Lines cut by the role boundary (HT * RECB not on a line) leave as
// the part this role owns.  Needs trajectories that start on a line: (S * RECB) % 128 == 0.  All of it is compile-time
// arithmetic once the segment loop is unrolled.
template <int O, int S> struct LineGeom {
    static constexpr int RECB = 48 * O;                       // bytes per (trajectory, segment) record
    static constexpr bool OK = (S * RECB) % 128 == 0;
    // bytes waiting for their line are a multiple of gcd(RECB, 128) below 128: 96 / 112 / 64 / 112 at orders 2 / 3 / 4 / 5
    static constexpr int GCD = RECB % 128 == 0 ? 128 : (RECB % 64 == 0 ? 64 : (RECB % 32 == 0 ? 32 : 16));
    static constexpr int RINGB = ((RECB + 128 - GCD + 127) / 128) * 128;
    static constexpr int ROW = RINGB / 8 + 2;                 // doubles; stride in dwords = odd multiple of 4
    static constexpr int MAXL = RECB / 128 + 1;               // lines one record can complete
};
template <int O, int S, bool BOTTOM> struct LineRing {
    using G = LineGeom<O, S>;
    static constexpr int RECB = G::RECB, HT = (S + 1) / 2;
    static constexpr int LO = BOTTOM ? HT * RECB : 0, HI = BOTTOM ? S * RECB : HT * RECB;
    static constexpr int RS = S * RECB;                        // bytes between consecutive trajectories
    static constexpr int LINES = BOTTOM ? (HI / 128 - LO / 128) : (HI + 127) / 128;   // lines (whole or cut) of this role
    // ring position, in doubles, of trajectory byte x
    __device__ static __forceinline__ constexpr int pos(int x) { return (x % G::RINGB) / 8; }
    // Stores the lines completed by record g (already in the ring).  tbase: the slice's first trajectory in coeffs
    // (wave-uniform).  PRED: rows may be dead (ragged slice, skip mask): bit i of live8 = row i*8 + lane/8 is stored.
    template <bool PRED>
    __device__ static __forceinline__ void flush(int g, const double *stage, char *tbase, int lane, bool nt, unsigned live8) {
        const int q = lane >> 3, p = lane & 7;
        const int l_lane = q * G::ROW + p * 2;
        const unsigned g_lane = (unsigned)(q * RS + p * 16);
        const int rlo = g * RECB, rhi = rlo + RECB;
        // top role (written = [rlo, HI)): lines that START inside the record; bottom role (written = [LO, rhi)): lines that END inside it
        const int l0 = BOTTOM ? rlo / 128 : (rlo + 127) / 128;
        const int nl = BOTTOM ? rhi / 128 - rlo / 128 : (rhi + 127) / 128 - (rlo + 127) / 128;
#pragma unroll
        for (int k = 0; k < G::MAXL; ++k) {
            if (k < nl) {
                const int ell = l0 + k;
                const bool whole = BOTTOM ? ell * 128 >= LO : ell * 128 + 128 <= HI;
                const bool pv = whole || (BOTTOM ? ell * 128 + p * 16 >= LO : ell * 128 + p * 16 < HI);
                double2 v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const double2 *>(stage + l_lane + pos(ell * 128) + i * 8 * G::ROW);
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (pv && (!PRED || ((live8 >> i) & 1u))) store16(tbase + ell * 128 + (size_t)i * 8 * RS + g_lane, v[i], nt);
            }
        }
    }
};

// LDS geometry of one workgroup (64 trajectories, two waves)
template <int O, int S> struct FixedLds {
    static constexpr int HT = (S + 1) / 2, HB = S / 2;  // segments of the top / bottom role
    static constexpr int REC = 6 * O;                // doubles per (trajectory, segment) record
    static constexpr int WP_ROW = (S + 1) * 3;       // doubles per trajectory, unpadded (bank-clean for b64 reads)
    static constexpr int TM_ROW = S;                 // doubles per trajectory, unpadded (LDS-DMA and the 16-byte
                                                     // copy-in write linearly; the few time reads tolerate conflicts)
    // staging row: the record (+ for order 4 the 8 doubles held over from the pair's other record),
    // padded so that the row stride in dwords is an odd multiple of 4 (conflict-free ds_write_b128)
    // (whole-line rings, LineGeom, where the trajectories start on lines)
    static constexpr int STAGE_ROW = LineGeom<O, S>::OK ? LineGeom<O, S>::ROW : (O == 4 ? 34 : (O == 2 ? 14 : REC));
    static constexpr int WP_DOUBLES = 64 * WP_ROW;
    static constexpr int TM_DOUBLES = 64 * TM_ROW;
    static constexpr int STAGE_DOUBLES = 64 * STAGE_ROW;  // per wave
    // the Schur carries of the exchange step live at the start of the PARTNER's staging tile
    // (written before the exchange barrier, read after it, before the tile is used for output)
    static constexpr int CARRY = (O - 1) * O / 2 + 3 * (O - 1);
    static_assert(CARRY * 64 <= STAGE_DOUBLES, "carries must fit in a staging tile");
    static constexpr int TOTAL_DOUBLES = WP_DOUBLES + TM_DOUBLES + 2 * STAGE_DOUBLES;
    // single-record store burst: LPR lanes per record (16-byte pieces), RPI records per wave store
    static constexpr int LPR = 3 * O;
    static constexpr int RPI = 64 / LPR;
    static constexpr int NI = (64 + RPI - 1) / RPI;
};

struct NoHook { __device__ __forceinline__ void operator()() const {} };

// Input accessor: local (role-oriented) segment times T(j), j = 0..HS-1, and waypoints P(j, axis),
// j = 0..HS, read on demand from the workgroup's LDS image.  The bottom role walks backwards.
template <int S, bool BOTTOM> struct LdsInputs {
    const double *l_wp, *l_tm;
    int lane;
    __device__ __forceinline__ double T(int j) const { return l_tm[lane * S + (BOTTOM ? S - 1 - j : j)]; }
    __device__ __forceinline__ double P(int j, int ax) const { return l_wp[lane * (S + 1) * 3 + (BOTTOM ? S - j : j) * 3 + ax]; }
};

// STASH: the forward sweep keeps the times/waypoints it reads in registers for the backward sweep,
// so the LDS input image is dead after the exchange barrier; `after_exchange()` runs right after
// that barrier (the persistent kernel issues the next slice's LDS-DMA there).
// This role's boundary derivatives in its own orientation (minimum_snap.cpp:527-555): velocity
// (order >= 2) and acceleration (order >= 3) are given, every higher one is pinned to 0; time
// reversal negates odd derivatives.  Loaded ONCE and kept in registers: re-reading them after the
// store bursts would force a vmcnt wait that drains every store of the slice.
template <bool BOTTOM> struct RoleBc {
    double v[3], acc[3], vw;
    __device__ __forceinline__ void load(const GenericArgs &a, int64_t b) {
        const double *bc = (const double *)a.bc + (a.bc_per_traj ? b * 12 : 0);
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
            v[ax] = BOTTOM ? -bc[1 * 3 + ax] : bc[0 * 3 + ax];
            acc[ax] = BOTTOM ? bc[3 * 3 + ax] : bc[2 * 3 + ax];
        }
        vw = a.vw_per ? a.vw_per[b] : a.vel_zero_weight;
    }
    __device__ __forceinline__ double at(int r, int ax) const { return r == 0 ? v[ax] : r == 1 ? acc[ax] : 0.0; }
    // the same for an axis index that differs between lanes (selects instead of a dynamically indexed register array)
    __device__ __forceinline__ double at_sel(int r, int ax) const {
        return r == 0 ? (ax == 0 ? v[0] : ax == 1 ? v[1] : v[2]) : r == 1 ? (ax == 0 ? acc[0] : ax == 1 ? acc[1] : acc[2]) : 0.0;
    }
    // batch-wide values are wave-uniform: park them in scalar registers for the life of the slice loop
    __device__ __forceinline__ void to_sgpr() {
        auto u = [](double x) {
            const unsigned long long q = __builtin_bit_cast(unsigned long long, x);
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)q), hi = __builtin_amdgcn_readfirstlane((unsigned)(q >> 32));
            return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
        };
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) { v[ax] = u(v[ax]); acc[ax] = u(acc[ax]); }
        vw = u(vw);
    }
};

// NAX = 3: one lane per trajectory (all three axes).  NAX = 1 (small batches, see launch_s): THREE lanes per trajectory,
// lane = (axis ax0, staging row `row`), each factorising redundantly and carrying one right-hand side -- about half the
// instructions per lane, which is what a lone latency-bound wave is made of.
template <int O, int S, bool BOTTOM, bool STATUS, bool FULL, bool SEGMAJ, bool STASH, class In, class Hook, int NAX = 3>
__device__ __forceinline__ void fixed_body(const GenericArgs &a, int64_t b0, int64_t b, int lane,
                                           const In &in, const RoleBc<BOTTOM> &rbc, double *stage, double *partner_stage,
                                           double *tst, const Hook &after_exchange, int rows = 64, int row_ = 0, int ax0 = 0) {
    static_assert(NAX == 3 || (NAX == 1 && !FULL && !STASH), "the axis-per-lane mapping serves the narrow one-slice kernel only");
    const int row = NAX == 3 ? lane : row_;   // staging-tile row = trajectory within the slice
    constexpr int N = O - 1, M = 2 * O;
    constexpr int HS = BOTTOM ? S / 2 : (S + 1) / 2;   // segments of THIS role; both roles meet at waypoint ceil(S/2)
    using L = FixedLds<O, S>;
    auto Tl = [&](int j) { return in.T(j); };
    auto Pl = [&](int j, int ax) { return in.P(j, ax0 + ax); };
    const double vw = rbc.vw;
    auto bc_at = [&](int r, int ax) { return NAX == 3 ? rbc.at(r, ax) : rbc.at_sel(r, ax0 + ax); };

    double z[N][NAX], W[N][N];
#pragma unroll
    for (int r = 0; r < N; ++r) {
#pragma unroll
        for (int ax = 0; ax < NAX; ++ax) z[r][ax] = bc_at(r, ax);
#pragma unroll
        for (int c = 0; c < N; ++c) W[r][c] = 0.0;
    }
    double Wst[HS][N][N], zst[HS][N][NAX];  // slot k = local waypoint k (slot 0 unused)
    bool spd = true;

    // ---- forward elimination over local interior waypoints 1..HS-1 ----
    Seg<O> left, right;
    double Pst[HS + 1][NAX];  // STASH only (the segment times are stashed in LDS: tst[j*64 + lane])
    { const double t0 = Tl(0); if (STASH) tst[lane] = t0; seg_make<O>(t0, vw, left); }
    double Pa[NAX], Pb[NAX], Pc[NAX];  // local waypoints k-1, k, k+1
#pragma unroll
    for (int ax = 0; ax < NAX; ++ax) {
        Pa[ax] = Pl(0, ax);
        Pb[ax] = Pl(1, ax);
        if (STASH) { Pst[0][ax] = Pa[ax]; Pst[1][ax] = Pb[ax]; }
    }
#pragma unroll
    for (int k = 1; k < HS; ++k) {
        { const double tk = Tl(k); if (STASH) tst[k * 64 + lane] = tk; seg_make<O>(tk, vw, right); }
#pragma unroll
        for (int ax = 0; ax < NAX; ++ax) { Pc[ax] = Pl(k + 1, ax); if (STASH) Pst[k + 1][ax] = Pc[ax]; }
        double Sm[N][N], R[N][N + NAX];  // right-hand sides: [C_k | y_k]
#pragma unroll
        for (int r = 0; r < N; ++r) {
#pragma unroll
            for (int c = 0; c <= r; ++c) {
                double v = ee_of<O>(left, r, c) + right.ss[r][c];
#pragma unroll
                for (int j = 0; j < N; ++j) v = __builtin_fma(-left.se[j][r], W[j][c], v);
                Sm[r][c] = v;
            }
#pragma unroll
            for (int c = 0; c < N; ++c) R[r][c] = right.se[r][c];
#pragma unroll
            for (int ax = 0; ax < NAX; ++ax) {
                double v = left.ep[r] * (Pb[ax] - Pa[ax]);
                v = __builtin_fma(right.sp[r], Pc[ax] - Pb[ax], v);
#pragma unroll
                for (int j = 0; j < N; ++j) v = __builtin_fma(-left.se[j][r], z[j][ax], v);
                R[r][N + ax] = v;
            }
        }
        spd &= SmallSpd<N, N + NAX>::solve(Sm, R);
#pragma unroll
        for (int r = 0; r < N; ++r) {
#pragma unroll
            for (int c = 0; c < N; ++c) { W[r][c] = R[r][c]; Wst[k][r][c] = R[r][c]; }
#pragma unroll
            for (int ax = 0; ax < NAX; ++ax) { z[r][ax] = R[r][N + ax]; zst[k][r][ax] = R[r][N + ax]; }
        }
        left = right;
#pragma unroll
        for (int ax = 0; ax < NAX; ++ax) { Pa[ax] = Pb[ax]; Pb[ax] = Pc[ax]; }
    }

    CSP_STAMP(2);
    // ---- Schur carry of this half onto the middle waypoint, exchanged through LDS ----
    double Cm[N][N], cm[N][NAX];
#pragma unroll
    for (int r = 0; r < N; ++r) {
#pragma unroll
        for (int c = 0; c <= r; ++c) {
            double v = ee_of<O>(left, r, c);
#pragma unroll
            for (int j = 0; j < N; ++j) v = __builtin_fma(-left.se[j][r], W[j][c], v);
            Cm[r][c] = v;
        }
#pragma unroll
        for (int ax = 0; ax < NAX; ++ax) {
            double v = left.ep[r] * (Pb[ax] - Pa[ax]);
#pragma unroll
            for (int j = 0; j < N; ++j) v = __builtin_fma(-left.se[j][r], z[j][ax], v);
            cm[r][ax] = v;
        }
    }
    {
        double *mine = partner_stage;  // the partner reads it from ITS tile after the barrier
        int e = 0;
#pragma unroll
        for (int r = 0; r < N; ++r)
#pragma unroll
            for (int c = 0; c <= r; ++c) mine[(e++) * 64 + lane] = Cm[r][c];
#pragma unroll
        for (int r = 0; r < N; ++r)
#pragma unroll
            for (int ax = 0; ax < NAX; ++ax) mine[(e++) * 64 + lane] = cm[r][ax];
    }
    lds_barrier();
    after_exchange();
    CSP_STAMP(3);
    double xm[N][NAX];
    {
        const double *other = stage;
        double Sm[N][N], R[N][NAX];
        int e = 0;
#pragma unroll
        for (int r = 0; r < N; ++r)
#pragma unroll
            for (int c = 0; c <= r; ++c) {
                const double o = other[(e++) * 64 + lane];
                Sm[r][c] = Cm[r][c] + (((r + c) & 1) ? -o : o);   // the other side's carry, conjugated by the
            }                                                      // odd-derivative sign flip
#pragma unroll
        for (int r = 0; r < N; ++r)
#pragma unroll
            for (int ax = 0; ax < NAX; ++ax) {
                const double o = other[(e++) * 64 + lane];
                R[r][ax] = cm[r][ax] + ((r & 1) ? o : -o);  // derivative r+1 is odd for even r
            }
        spd &= SmallSpd<N, NAX>::solve(Sm, R);
#pragma unroll
        for (int r = 0; r < N; ++r)
#pragma unroll
            for (int ax = 0; ax < NAX; ++ax) xm[r][ax] = R[r][ax];
    }

    // ---- back-substitution fused with coefficient recovery, local segments HS-1 .. 0 ----
    double nanacc = 0.0;
    // bytes between consecutive trajectories' records of one segment: the default layout is
    // [B][S][3][2o]; CSP_FLAG_SEGMENT_MAJOR selects [S][B][3][2o]
    constexpr int RECB = L::REC * 8;
    constexpr int RS = SEGMAJ ? RECB : S * RECB;
    constexpr int ROW = L::STAGE_ROW;
    // Output leaves through a lane-major LDS tile and is read back transposed.  For order 4 in the
    // default layout the records of segments (2q, 2q+1) of one trajectory form one 384-byte,
    // 128-byte-aligned run, so a wave pairs them: of the first record it stores the 128 bytes that
    // complete a cache line and holds the other 64 in the tile; with the second record it stores a
    // 256-byte run.  Every line is then written whole (the single-record scheme left 1/3 of the lines
    // half-written between two bursts and measured +8 % WRITE_SIZE).  Lane maps of the burst shapes:
    constexpr bool PAIRING = FULL && !SEGMAJ && O == 4 && (S % 2) == 0;
    // the other orders: whole-line ring (LineRing above) wherever the trajectories start on 128-byte lines
    constexpr bool RING = FULL && !SEGMAJ && !PAIRING && NAX == 3 && LineGeom<O, S>::OK;
    using LR = LineRing<O, S, BOTTOM>;
    const int grp = lane / L::LPR;                 // LPR lanes per record (lanes >= RPI*LPR idle)
    const int lane_in = lane - grp * L::LPR;
    const int lds_off = grp * ROW + lane_in * 2;   // doubles
    const unsigned g_off = (unsigned)(grp * RS + lane_in * 16);  // bytes
    const int l8 = (lane >> 3) * ROW + 8 + (lane & 7) * 2;       // 8 lanes per 128-byte half, tile doubles 8..23
    const unsigned o8 = (unsigned)((lane >> 3) * RS + (lane & 7) * 16);
    const int l16 = (lane >> 4) * ROW + (lane & 15) * 2;         // 16 lanes per 256-byte run, tile doubles 0..31
    const unsigned o16 = (unsigned)((lane >> 4) * RS + (lane & 15) * 16);
    double xn[N][NAX];  // free derivatives at local waypoint j+1
#pragma unroll
    for (int r = 0; r < N; ++r)
#pragma unroll
        for (int ax = 0; ax < NAX; ++ax) xn[r][ax] = xm[r][ax];
#pragma unroll
    for (int j = HS - 1; j >= 0; --j) {
        double xk[N][NAX];  // free derivatives at local waypoint j
#pragma unroll
        for (int r = 0; r < N; ++r)
#pragma unroll
            for (int ax = 0; ax < NAX; ++ax) {
                if (j == 0) {
                    xk[r][ax] = bc_at(r, ax);
                } else {
                    double v = zst[j][r][ax];
#pragma unroll
                    for (int c = 0; c < N; ++c) v = __builtin_fma(-Wst[j][r][c], xn[c][ax], v);
                    xk[r][ax] = v;
                }
            }
        const double Tj = STASH ? tst[j * 64 + lane] : Tl(j);
        double ip[M], tp[N];
        ip[0] = 1.0;
        ip[1] = fast_rcp(Tj);
#pragma unroll
        for (int e = 2; e < M; ++e) ip[e] = ip[e - 1] * ip[1];
        tp[0] = Tj;
#pragma unroll
        for (int e = 1; e < N; ++e) tp[e] = tp[e - 1] * Tj;
        const int g = BOTTOM ? S - 1 - j : j;  // global segment index
        // the middle pair of an odd half is split between the two waves: those records go out singly
        const bool paired = PAIRING && !((HS & 1) && g == (BOTTOM ? HS : HS - 1));
        const bool first = BOTTOM ? (g & 1) == 0 : (g & 1) == 1;  // first record of its pair to reach this wave
#pragma unroll
        for (int ax = 0; ax < NAX; ++ax) {
            double xs[N], xe[N], c[M];
            // global orientation: the bottom role's local start is the global END, and odd
            // derivatives change sign back
#pragma unroll
            for (int r = 0; r < N; ++r) {
                const double sgn = (BOTTOM && !(r & 1)) ? -1.0 : 1.0;
                xs[r] = BOTTOM ? sgn * xn[r][ax] : xk[r][ax];
                xe[r] = BOTTOM ? sgn * xk[r][ax] : xn[r][ax];
            }
            const double Plo = STASH ? Pst[j][ax] : Pl(j, ax), Phi = STASH ? Pst[j + 1][ax] : Pl(j + 1, ax);
            const double Ps = BOTTOM ? Phi : Plo;
            const double Pe = BOTTOM ? Plo : Phi;
            recover<O>(Ps, Pe - Ps, xs, xe, tp, ip, c);
            // lane-major staging tile (row = lane); where the axis block lands depends on the
            // record's place in its pair (order 4 only, see below)
            const int tpos = !paired ? (ax0 + ax) * M
                           : (!BOTTOM ? (first ? (ax == 0 ? 24 : ax * M) : ax * M)
                                      : (first ? (ax == 2 ? 0 : 8 + ax * M) : 8 + ax * M));
#pragma unroll
            for (int i = 0; i < M; i += 2) {
                double2 v2;
                v2.x = c[i];
                v2.y = c[i + 1];
                const int at = RING ? LR::pos(g * RECB + ((ax0 + ax) * M + i) * 8) : tpos + i;
                *reinterpret_cast<double2 *>(stage + row * ROW + at) = v2;
            }
            if (STATUS) {
                // Non-finite values are caught on the highest-power and the constant coefficient: every endpoint
                // quantity of the segment (dP, the 2(o-1) scaled derivatives) enters c[0] with a non-zero weight
                // G(0, .) (tests/test_tables.py) times T^-(2o-1), and c[M-1] is the start waypoint, so a NaN/Inf
                // anywhere in the inputs or the unknowns reaches one of the two.  Testing all 2o coefficients kept
                // them live across the staging writes (0.35 KB of scratch per lane at order 4, S >= 15).
                nanacc = __builtin_fma(c[0], 0.0, nanacc);
                nanacc = __builtin_fma(c[M - 1], 0.0, nanacc);
            }
        }
        // LDS operations of one wave execute in order, so the tile needs no barrier; the fences only
        // stop the compiler from reordering the (may-alias) LDS accesses.
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        if (paired) {
            // pair base = record of the even segment; TOP meets the odd record first, BOTTOM the even one
            char *pbase = reinterpret_cast<char *>((double *)a.coeffs + (b0 * S + (g & ~1)) * L::REC);  // uniform
            const bool nt = a.nt_stores != 0;
            if (first) {
                double2 v[8];   // 8 rows x 128 bytes per store
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const double2 *>(stage + l8 + i * 8 * ROW);
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    store16(pbase + (BOTTOM ? 0 : 256) + (size_t)i * 8 * RS + o8, v[i], nt);
            } else {
#pragma unroll
                for (int h = 0; h < 2; ++h) {  // 4 rows x 256 bytes per store, two batches of 8
                    double2 v[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const double2 *>(stage + l16 + (h * 8 + i) * 4 * ROW);
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        store16(pbase + (BOTTOM ? 128 : 0) + (size_t)(h * 8 + i) * 4 * RS + o16, v[i], nt);
                }
            }
        } else if (RING) {
            LR::template flush<false>(g, stage, reinterpret_cast<char *>((double *)a.coeffs + b0 * S * L::REC), lane, a.nt_stores != 0, 0xffu);
        } else {
            char *gbase = reinterpret_cast<char *>((double *)a.coeffs + (SEGMAJ ? ((int64_t)g * a.Btotal + a.Boffset + b0) : (b0 * S + g)) * L::REC);  // uniform
            if (FULL) {
                // NI stores of RPI whole records each; lanes beyond RPI*LPR and rows beyond 63 are masked
                // off, not made to repeat a neighbour's piece (duplicates are traffic)
                if (lane < L::RPI * L::LPR) {
                    constexpr int NFULL = 64 / L::RPI;      // stores whose RPI rows are all < 64
                    double2 v[NFULL];
#pragma unroll
                    for (int i = 0; i < NFULL; ++i)
                        v[i] = *reinterpret_cast<const double2 *>(stage + lds_off + i * L::RPI * ROW);
                    const bool nt = a.nt_stores != 0;
#pragma unroll
                    for (int i = 0; i < NFULL; ++i) store16(gbase + (size_t)i * L::RPI * RS + g_off, v[i], nt);
                    if (NFULL < L::NI && NFULL * L::RPI + grp < 64)   // the ragged last store
                        store16(gbase + (size_t)NFULL * L::RPI * RS + g_off,
                                *reinterpret_cast<const double2 *>(stage + lds_off + NFULL * L::RPI * ROW), nt);
                }
            } else {
                // the axis-per-lane mapping serves slices of <= 16 rows: only the first stores can hold one
                constexpr int NI_USED = NAX == 3 ? L::NI : (16 + L::RPI - 1) / L::RPI;
#pragma unroll
                for (int i = 0; i < NI_USED; ++i) {
                    const int srow = i * L::RPI + grp;
                    if (lane < L::RPI * L::LPR && srow < rows) {
                        const double2 v2 = *reinterpret_cast<const double2 *>(stage + lds_off + i * L::RPI * ROW);
                        *reinterpret_cast<double2 *>(gbase + (size_t)i * L::RPI * RS + g_off) = v2;
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
        for (int r = 0; r < N; ++r)
#pragma unroll
            for (int ax = 0; ax < NAX; ++ax) xn[r][ax] = xk[r][ax];
    }
    CSP_STAMP(4);
    CSP_STAMP_RT(6);
    if (STATUS && row < rows) {
        const int bits = (spd ? 0 : 2) | ((nanacc == 0.0) ? 0 : 1);
        if (bits) atomicOr(a.status + b, bits);
    }
}

// FULL = every workgroup owns 64 real trajectories (B % 64 == 0); the ragged remainder of a
// batch is a second, single-workgroup launch of the FULL=false variant.
template <int O, int S, bool STATUS, bool FULL, bool SEGMAJ, int NAX = 3>
__global__ void __launch_bounds__(128) minsnap_fixed_kernel(GenericArgs a, MultiTable mt) {
    using L = FixedLds<O, S>;
    __shared__ __attribute__((aligned(16))) double lds[L::TOTAL_DOUBLES];
    double *l_wp = lds;
    double *l_tm = l_wp + L::WP_DOUBLES;
    double *l_stage = l_tm + L::TM_DOUBLES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int role = tid >> 6;  // wave-uniform
    // FULL: 64 trajectories per workgroup; otherwise a.slice_w (<= 64) of them -- the ragged tail of a batch, or a small
    // batch cut into narrow slices (lanes >= rows compute on an unloaded image and store nothing)
    const int slice_w = FULL ? 64 : a.slice_w;
    int64_t blk = blockIdx.x;
    if (!FULL && mt.n) {
        // csp_minsnap_solve_multi: this workgroup's batch (scalar search of the kernel-argument table), whose buffers replace
        // the launch-wide ones
        int k = 0;
        while (k + 1 < mt.n && (int)blockIdx.x >= mt.first_slice[k + 1]) ++k;
        blk -= mt.first_slice[k];
        a.wp = mt.e[k].wp; a.times = mt.e[k].tm; a.bc = mt.e[k].bc; a.coeffs = mt.e[k].co; a.status = mt.e[k].status; a.B = mt.e[k].B;
    }
    const int64_t b0 = blk * slice_w;
    const int rows = (int)((a.B - b0) < slice_w ? (a.B - b0) : slice_w);
    if (!FULL && STATUS && mt.n) {
        // one launch, many status arrays: cleared here instead of by a memset per batch (the __syncthreads below orders
        // these stores before the atomicOr of either role)
        if (role == 0 && lane < rows && NAX == 3) a.status[b0 + lane] = 0;
    }

    CSP_STAMP_RT(5);
    CSP_STAMP(0);
    // ---- coalesced copy-in: the workgroup's waypoints and times are contiguous in HBM ----
    {
        const double2 *g_wp = reinterpret_cast<const double2 *>((const double *)a.wp + b0 * L::WP_ROW);
        const int n_wp = rows * L::WP_ROW / 2;  // 16-byte pieces
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
        if ((rows * L::WP_ROW) & 1) {  // odd number of doubles: last one by itself
            if (tid == 0) l_wp[rows * L::WP_ROW - 1] = ((const double *)a.wp + b0 * L::WP_ROW)[rows * L::WP_ROW - 1];
        }
        const double2 *g_tm = reinterpret_cast<const double2 *>((const double *)a.times + b0 * S);
        const int n_tm = rows * S / 2;  // 16-byte pieces
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
        if ((rows * S) & 1) {
            if (tid == 0) l_tm[rows * S - 1] = ((const double *)a.times + b0 * S)[rows * S - 1];
        }
    }
    __syncthreads();
    CSP_STAMP(1);

    // NAX = 1 (slices of <= 16 trajectories): lane = (axis, row); the lanes beyond 3 * slice_w repeat axis 2 (same values
    // into the same staging cells)
    const int row = NAX == 3 ? lane : (lane & (slice_w - 1));
    const int ax0 = NAX == 3 ? 0 : ((lane / slice_w) < 2 ? (lane / slice_w) : 2);
    int64_t b = b0 + row;
    if (b >= a.B) b = a.B - 1;  // idle lanes of a ragged last workgroup: harmless, store nothing
    if (role == 0) {
        const LdsInputs<S, false> in{l_wp, l_tm, row};
        RoleBc<false> rbc;
        rbc.load(a, b);
        fixed_body<O, S, false, STATUS, FULL, SEGMAJ, false, LdsInputs<S, false>, NoHook, NAX>(
            a, b0, b, lane, in, rbc, l_stage, l_stage + L::STAGE_DOUBLES, nullptr, NoHook{}, rows, row, ax0);
    } else {
        const LdsInputs<S, true> in{l_wp, l_tm, row};
        RoleBc<true> rbc;
        rbc.load(a, b);
        fixed_body<O, S, true, STATUS, FULL, SEGMAJ, false, LdsInputs<S, true>, NoHook, NAX>(
            a, b0, b, lane, in, rbc, l_stage + L::STAGE_DOUBLES, l_stage, nullptr, NoHook{}, rows, row, ax0);
    }
}

// LDS-DMA (global_load_lds_dwordx4: HBM -> LDS, no registers in between) of one 64-trajectory
// slice: waypoints then times, copied linearly in 16-byte pieces, 1 KiB per wave instruction.
template <int S> struct SlicePrefetch {
    typedef const __attribute__((address_space(1))) void *gptr_t;
    typedef __attribute__((address_space(3))) void *lptr_t;
    typedef __attribute__((address_space(3))) char *lchar_t;
    static constexpr int WP_ROW = (S + 1) * 3;
    static constexpr int WP_PIECES = 64 * WP_ROW / 2;   // 64*WP_ROW is even
    static constexpr int TM_PIECES = 64 * S / 2;
    static constexpr int WP_ITERS = (WP_PIECES + 127) / 128, TM_ITERS = (TM_PIECES + 127) / 128;
    static constexpr int TM_BYTE_OFF = 64 * WP_ROW * 8;
    const char *wp, *tm;     // batch base pointers
    lchar_t lds3;            // LDS image base (waypoints, then unpadded times)
    int tid, role;
    int64_t next, n_slices;
    // One 16-byte piece per lane, HBM -> LDS at (wave-uniform base + lane*16).  Written as inline asm
    // on purpose: hipcc's waitcnt insertion answers a pending LDS-DMA with `s_waitcnt vmcnt(0)` in
    // front of the first LDS read that may alias it, which here would drain the ~100 stores issued
    // after the prefetch.  The kernel instead waits with a COUNTED vmcnt at the top of the slice loop
    // (persistent_role_loop) -- this statement is the only place that wait protects.
    __device__ __forceinline__ void dma16(const char *src, int lds_byte_off) const {
        const unsigned m0v = (unsigned)(size_t)lds3 + (unsigned)lds_byte_off;  // LDS byte address, wave-uniform
        asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off"
                     :: "v"(src), "s"(__builtin_amdgcn_readfirstlane(m0v)) : "memory");  // m0 is a reserved register: hipcc neither tracks nor relies on it here
                                      // (gfx950 DS instructions take no m0; this kernel has no other m0 user)
    }
    __device__ __forceinline__ void issue(int64_t slice) const {
        const char *g_wp = wp + slice * (64 * WP_ROW * 8);
        const char *g_tm = tm + slice * (64 * S * 8);
#pragma unroll
        for (int it = 0; it < WP_ITERS; ++it) {
            const int q = it * 128 + tid;  // piece index; a wave's 64 pieces are contiguous
            if (q < WP_PIECES) dma16(g_wp + (size_t)q * 16, (it * 128 + role * 64) * 16);
        }
#pragma unroll
        for (int it = 0; it < TM_ITERS; ++it) {
            const int q = it * 128 + tid;
            if (q < TM_PIECES) dma16(g_tm + (size_t)q * 16, TM_BYTE_OFF + (it * 128 + role * 64) * 16);
        }
    }
    __device__ __forceinline__ void operator()() const { if (next < n_slices) issue(next); }
};

template <int O, int S, bool BOTTOM, bool STATUS, bool SEGMAJ>
__device__ __forceinline__ void persistent_role_loop(const GenericArgs &a, int n_slices, int lane, const double *l_wp,
                                                     const double *l_tm, double *stage, double *partner_stage,
                                                     double *tst, SlicePrefetch<S> pf) {
    constexpr int HS = BOTTOM ? S / 2 : (S + 1) / 2;
    using L = FixedLds<O, S>;
    // Vector-memory operations a wave issues AFTER a slice's prefetch and before the next top-of-loop
    // wait: its store bursts.  Must not be over-estimated (the counted wait below relies on at least
    // this many younger operations existing).  Paired records (order 4, default layout): 8 + 16
    // stores per pair; single records: NI each.
    constexpr int PAIRS = (SEGMAJ || O != 4 || (S % 2) != 0) ? 0 : HS / 2;
    constexpr bool RING = !SEGMAJ && O != 4 && LineGeom<O, S>::OK;   // as in fixed_body: 8 stores per line of the role
    constexpr int STORES_PER_SLICE = RING ? LineRing<O, S, BOTTOM>::LINES * 8 : PAIRS * 24 + (HS - 2 * PAIRS) * L::NI;
    // Batch-wide boundary conditions / weight are read once, before the loop: a global load inside
    // the loop could only be waited for together with every older store.  (Per-trajectory boundary
    // conditions or weights take the one-workgroup-per-slice kernel instead, see launch_hs.)
    RoleBc<BOTTOM> rbc;
    rbc.load(a, 0);
    rbc.to_sgpr();
    bool first = true;
    for (int64_t slice = blockIdx.x; slice < n_slices; slice += gridDim.x) {
        // the prefetch of this slice is older than every store of the previous slice, so waiting for
        // all but the youngest min(63, stores) operations covers it without draining the stores
        if (first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(STORES_PER_SLICE < 63 ? STORES_PER_SLICE : 63) : "memory");
        lds_barrier();
        if (first) CSP_STAMP(1);
        const LdsInputs<S, BOTTOM> in{l_wp, l_tm, lane};
        const int64_t b0 = slice * 64;
        pf.next = slice + gridDim.x;
        // the image is dead once both waves passed the exchange barrier: prefetch the next slice there
        fixed_body<O, S, BOTTOM, STATUS, true, SEGMAJ, true>(a, b0, b0 + lane, lane, in, rbc, stage, partner_stage, tst, pf);
        first = false;
    }
}

// Persistent variant for the full workgroups of a batch: gridDim.x workgroups (two per CU) walk the
// batch with stride gridDim.x; the NEXT slice's inputs stream into LDS while the current slice is
// back-substituted and stored, so only a workgroup's very first copy-in is exposed.
template <int O, int S, bool STATUS, bool SEGMAJ>
__global__ void __launch_bounds__(128) minsnap_fixed_persistent_kernel(GenericArgs a, int n_slices) {
    using L = FixedLds<O, S>;
    // image (waypoints, times) | two staging tiles | two time stashes (the forward sweep parks the
    // segment times it read there: the image is overwritten by the prefetch during the backward sweep)
    __shared__ __attribute__((aligned(16))) double lds[64 * L::WP_ROW + 64 * S + 2 * L::STAGE_DOUBLES + 2 * L::HT * 64];
    double *l_wp = lds;
    double *l_tm = l_wp + 64 * L::WP_ROW;
    double *l_stage = l_tm + 64 * S;
    double *l_tst = l_stage + 2 * L::STAGE_DOUBLES;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int role = tid >> 6;  // wave-uniform
    SlicePrefetch<S> pf;
    pf.wp = reinterpret_cast<const char *>(a.wp);
    pf.tm = reinterpret_cast<const char *>(a.times);
    pf.lds3 = (typename SlicePrefetch<S>::lchar_t)lds;  // cast straight from the LDS object
    pf.tid = tid;
    pf.role = role;
    pf.next = 0;
    pf.n_slices = n_slices;
    CSP_STAMP_RT(5);
    CSP_STAMP(0);
    if ((int64_t)blockIdx.x < n_slices) pf.issue(blockIdx.x);
    // one loop per role: each wave's instruction stream holds a single specialisation
    if (role == 0) persistent_role_loop<O, S, false, STATUS, SEGMAJ>(a, n_slices, lane, l_wp, l_tm, l_stage, l_stage + L::STAGE_DOUBLES, l_tst, pf);
    else persistent_role_loop<O, S, true, STATUS, SEGMAJ>(a, n_slices, lane, l_wp, l_tm, l_stage + L::STAGE_DOUBLES, l_stage, l_tst + L::HT * 64, pf);
}

// ---- host side: launch one order's kernels -----------------------------------------------------
// Trajectories per workgroup for the one-workgroup-per-slice kernel.  A small batch is latency-bound -- a lone wave's
// instruction stream is the run time -- so what helps is fewer instructions per lane, not more workgroups: with the
// three-lanes-per-trajectory mapping (order 4) batches of B <= 32 * CUs trajectories run as
// slices of 16 (measured, B = 4096, S = 8: 8.2 us one lane per trajectory in slices of 64, 7.9 us in slices of 16,
// 6.1 us with three lanes per trajectory; B = 8192: 9.3 -> 8.8 us).  Everything else keeps 64.  CSP_SLICE_W overrides
// (tuning experiments: 8, 16, 32, 64).
inline int narrow_slice(int64_t B, int cus, bool axis_lanes) {
    static const int forced = [] { const char *e = std::getenv("CSP_SLICE_W"); return e ? std::atoi(e) : 0; }();
    if (forced == 8 || forced == 16 || forced == 32 || forced == 64) return forced;   // even: 16-byte pieces stay aligned
    return (axis_lanes && B <= 32 * (int64_t)cus) ? 16 : 64;
}
inline bool axis_lanes_enabled() {   // CSP_AXIS_LANES=0 switches the three-lanes-per-trajectory mapping off (A/B runs)
    static const bool on = [] { const char *e = std::getenv("CSP_AXIS_LANES"); return !(e && e[0] == '0'); }();
    return on;
}

// Non-temporal coefficient stores (store16) once the coefficients of the launch exceed the 256 MB Infinity Cache;
// CSP_NT_STORES=0 / 1 forces the choice (A/B runs).
inline int nt_forced() {
    static const int forced = [] { const char *e = std::getenv("CSP_NT_STORES"); return e ? (e[0] == '0' ? 0 : 1) : -1; }();
    return forced;
}
inline int nt_stores_for(int64_t B, int S, int O) {
    const int forced = nt_forced();
    if (forced >= 0) return forced;
    return (double)B * S * 6 * O * 8.0 > 256.0 * 1024 * 1024 ? 1 : 0;
}

template <int O, int S, bool SEGMAJ_OK>
hipError_t launch_s(const GenericArgs &a, int cus, hipStream_t st) {
    const dim3 block(128);
    if (a.multi) {
        // csp_minsnap_solve_multi: every batch cut into slices of 64 (one lane per trajectory), one workgroup per slice,
        // all of them in one grid
        GenericArgs t = a;
        t.slice_w = 64;
        t.multi = nullptr;
        const dim3 grid((unsigned)a.multi->first_slice[a.multi->n]);
        bool any_status = false;
        for (int k = 0; k < a.multi->n; ++k) any_status |= a.multi->e[k].status != nullptr;
        if (any_status) hipLaunchKernelGGL((minsnap_fixed_kernel<O, S, true, false, false>), grid, block, 0, st, t, *a.multi);
        else hipLaunchKernelGGL((minsnap_fixed_kernel<O, S, false, false, false>), grid, block, 0, st, t, *a.multi);
        return hipGetLastError();
    }
    const bool axis_ok = O == 4 && !a.seg_major && axis_lanes_enabled();
    const int w = narrow_slice(a.B, cus, axis_ok);
    if (w < 64 && !a.seg_major) {
        GenericArgs t = a;
        t.slice_w = w;
        const dim3 grid((unsigned)((a.B + w - 1) / w));
        if constexpr (O == 4) {
            // slices of <= 16 trajectories: three lanes per trajectory, one per axis (fixed_body NAX = 1)
            if (w <= 16 && axis_ok) {
                if (a.status) hipLaunchKernelGGL((minsnap_fixed_kernel<O, S, true, false, false, 1>), grid, block, 0, st, t, MultiTable{});
                else hipLaunchKernelGGL((minsnap_fixed_kernel<O, S, false, false, false, 1>), grid, block, 0, st, t, MultiTable{});
                return hipGetLastError();
            }
        }
        if (a.status) hipLaunchKernelGGL((minsnap_fixed_kernel<O, S, true, false, false>), grid, block, 0, st, t, MultiTable{});
        else hipLaunchKernelGGL((minsnap_fixed_kernel<O, S, false, false, false>), grid, block, 0, st, t, MultiTable{});
        return hipGetLastError();
    }
    const int64_t n_full = a.B / 64, rem = a.B % 64;
    const int64_t pgrid = n_full < 2 * (int64_t)cus ? n_full : 2 * (int64_t)cus;  // two workgroups per CU
    GenericArgs t = a;  // tail: the last B % 64 trajectories, one workgroup
    if (rem) {
        const int64_t off = n_full * 64;
        t.B = rem;
        t.wp = (const double *)a.wp + off * (a.S + 1) * 3;
        t.times = (const double *)a.times + off * a.S;
        if (a.seg_major) t.Boffset = off;   // segment-major records are addressed from the batch start
        else t.coeffs = (double *)a.coeffs + off * a.S * 6 * O;
        if (a.bc_per_traj) t.bc = (const double *)a.bc + off * 12;
        if (a.status) t.status = a.status + off;
        if (a.vw_per) t.vw_per = a.vw_per + off;
    }
    auto go = [&](auto status_tag, auto segmaj_tag) {
        constexpr bool ST = decltype(status_tag)::value, SM = decltype(segmaj_tag)::value;
        if (n_full) {
            GenericArgs f = a;
            f.B = n_full * 64;
            // Non-temporal stores need lines that leave whole: order 4 (paired 192-byte records) and, from round 3 on, every
            // order whose trajectories start on 128-byte lines (LineRing).  Record-at-a-time stores of 96- / 144- / 240-byte
            // records share lines between instructions, which the L2 merges for ordinary stores and not for non-temporal
            // ones (round 2, B = 524288: 261 / 392 / 312 us ordinary against 499 / 841 / 404 us non-temporal).
            constexpr bool WHOLE_LINES = !SM && (O == 4 || LineGeom<O, S>::OK);
            f.nt_stores = (WHOLE_LINES || O == 4) ? nt_stores_for(a.B, S, O) : (nt_forced() == 1 ? 1 : 0);
            if (a.persistent && !a.bc_per_traj && !a.vw_per) hipLaunchKernelGGL((minsnap_fixed_persistent_kernel<O, S, ST, SM>), dim3((unsigned)pgrid), block, 0, st, f, (int)n_full);
            else hipLaunchKernelGGL((minsnap_fixed_kernel<O, S, ST, true, SM>), dim3((unsigned)n_full), block, 0, st, f, MultiTable{});
        }
        if (rem) hipLaunchKernelGGL((minsnap_fixed_kernel<O, S, ST, false, SM>), dim3(1), block, 0, st, t, MultiTable{});
    };
    if (a.seg_major) {
        if constexpr (SEGMAJ_OK) {
            if (a.status) go(std::true_type{}, std::true_type{});
            else go(std::false_type{}, std::true_type{});
        } else {
            return hipErrorInvalidValue;
        }
    } else {
        if (a.status) go(std::true_type{}, std::false_type{});
        else go(std::false_type{}, std::false_type{});
    }
    return hipGetLastError();
}

}  // namespace fixedk
}  // namespace csp
