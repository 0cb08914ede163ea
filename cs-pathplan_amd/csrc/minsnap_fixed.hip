// minsnap_fixed.hip -- register-resident kernel for the headline buckets
// (order 4 = minimum snap, fp64, uniform even S <= 16, no path penalty).
//
// Mapping: a trajectory is split at its middle waypoint between two WAVES of one
// workgroup.  Wave 0 ("top") eliminates interior waypoints 1..S/2-1 downwards, wave 1
// ("bottom") runs the very same code on the time-reversed second half (reversed waypoint order,
// odd derivatives negated), i.e. a twisted block-LDL^T factorisation of the block-tridiagonal
// R_PP (minimum_snap.cpp:564-566).  Lane l of both waves owns trajectory 64*block+l, so every
// value a lane needs later (W_k = S_k^-1 C_k and z_k = S_k^-1 y_k, 18 doubles per waypoint) stays
// in ITS registers -- nothing is spilled to memory between the forward and the backward sweep.
// The two halves meet once, through 15 doubles per lane in LDS: each side's Schur carry onto
// the middle waypoint.  The roles are wave-uniform, so the only divergence is a scalar branch.
//
// Algorithmic HBM traffic per trajectory: 8*(3(S+1)+S) bytes in, 8*24*S bytes out
// (S=16: 536 + 3072 = 3608 B); no workspace.
#include "minsnap_device.h"
#include "minsnap_launch.h"

#include <type_traits>

#ifdef CSP_STAMPS
// Diagnostic build only (python cs-pathplan_amd/build.py --stamps): per-wave s_memtime stamps
// written to a buffer nothing else reads.  The shipped library contains none of this.
__device__ unsigned long long csp_g_stamps[8192 * 8];
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
extern "C" int csp_debug_read_stamps(unsigned long long *host, size_t n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(csp_g_stamps), n * sizeof(unsigned long long));
}
#else
#define CSP_STAMP(slot) do { } while (0)
#define CSP_STAMP_RT(slot) do { } while (0)
#endif

namespace csp {

namespace {

using T4 = Tab<4>;
constexpr int O4 = 4;
constexpr int M8 = 8;

// Scaled per-segment constants for order 4 (free derivatives r = 1..3 -> index r-1).
// ee[r][c] = (-1)^(r+c) ss[r][c] and Qt[.][end pos] = -Qt[.][start pos] (checked in
// tests/test_tables.py), so only ss, se and the two position columns are formed.
// Workgroup barrier that orders LDS traffic only.  __syncthreads() would also wait for every
// outstanding global store (vmcnt(0)); the persistent kernel keeps stores and the next
// workgroup's LDS-DMA in flight across its barriers.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

struct Seg4 {
    double ss[3][3];  // symmetric; full storage keeps the unrolled code simple
    double se[3][3];
    double sp[3];     // Qt[start r][start pos]
    double ep[3];     // Qt[end r][start pos]
};

__device__ __forceinline__ void seg4(double T, double vw, Seg4 &s) {
    double ip[M8];
    ip[0] = 1.0;
    ip[1] = fast_rcp(T);
#pragma unroll
    for (int e = 2; e < M8; ++e) ip[e] = ip[e - 1] * ip[1];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            s.ss[r][c] = T4::QT(r + 1, c + 1) * ip[5 - r - c];
            s.se[r][c] = T4::QT(r + 1, O4 + c + 1) * ip[5 - r - c];
        }
        s.sp[r] = T4::QT(r + 1, 0) * ip[6 - r];
        s.ep[r] = T4::QT(O4 + r + 1, 0) * ip[6 - r];
    }
    s.ss[0][0] += vw;  // zero-velocity penalty: +w on the velocity diagonal (minimum_snap.cpp:473-509)
}

__device__ __forceinline__ double ee_of(const Seg4 &s, int r, int c) {
    return ((r + c) & 1) ? -s.ss[r][c] : s.ss[r][c];
}

// Inverse of a symmetric positive definite 3x3 via cofactors and one reciprocal.
// Returns false when a leading minor is not positive.
__device__ __forceinline__ bool inv3(const double (&S)[3][3], double (&I)[3][3]) {
    const double a = S[0][0], b = S[1][0], c = S[1][1], d = S[2][0], e = S[2][1], f = S[2][2];
    const double c00 = __builtin_fma(c, f, -e * e);
    const double c10 = __builtin_fma(d, e, -b * f);
    const double c20 = __builtin_fma(b, e, -c * d);
    const double c11 = __builtin_fma(a, f, -d * d);
    const double c21 = __builtin_fma(b, d, -a * e);
    const double c22 = __builtin_fma(a, c, -b * b);
    const double det = __builtin_fma(a, c00, __builtin_fma(b, c10, d * c20));
    const double rd = fast_rcp(det);
    I[0][0] = c00 * rd;
    I[1][0] = I[0][1] = c10 * rd;
    I[2][0] = I[0][2] = c20 * rd;
    I[1][1] = c11 * rd;
    I[2][1] = I[1][2] = c21 * rd;
    I[2][2] = c22 * rd;
    return (a > 0.0) && (c22 > 0.0) && (det > 0.0);
}

// Hermite -> monomial map for the 4 high coefficients (t^7..t^4) of one segment and axis.
// xs/xe: free derivatives (vel, acc, jerk) at the segment's start/end in GLOBAL orientation,
// dP = P_end - P_start, tp[r] = T^(r+1), ip[e] = T^-e.
__device__ __forceinline__ void recover4(double Ps, double dP, const double (&xs)[3], const double (&xe)[3],
                                         const double (&tp)[3], const double (&ip)[M8], double (&c)[M8]) {
    double hs[3], he[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) { hs[r] = xs[r] * tp[r]; he[r] = xe[r] * tp[r]; }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        // G[i][0] + G[i][4] = 0 for the high rows: positions enter through dP only
        double acc = T4::G(i, O4) * dP;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            acc = __builtin_fma(T4::G(i, r + 1), hs[r], acc);
            acc = __builtin_fma(T4::G(i, O4 + r + 1), he[r], acc);
        }
        c[i] = acc * ip[7 - i];
    }
    c[4] = xs[2] * T4::G(4, 3);  // jerk / 3!
    c[5] = xs[1] * T4::G(5, 2);  // acc / 2!
    c[6] = xs[0];
    c[7] = Ps;
}

// LDS geometry of one workgroup (64 trajectories, two waves)
template <int HS> struct FixedLds {
    static constexpr int S = 2 * HS;
    static constexpr int WP_ROW = (S + 1) * 3;       // doubles per trajectory, unpadded (bank-clean for b64 reads)
    static constexpr int TM_ROW = S + 2;             // doubles per trajectory, padded against bank conflicts
    static constexpr int STAGE_ROW = 34;             // 24 coefficients + 8 held over from the pair's other
                                                     // record + 2 pad doubles (272 B rows: conflict-free b128)
    static constexpr int WP_DOUBLES = 64 * WP_ROW;
    static constexpr int TM_DOUBLES = 64 * TM_ROW;
    static constexpr int STAGE_DOUBLES = 64 * STAGE_ROW;  // per wave
    // the 15-double Schur carries of the exchange step live at the start of the PARTNER's staging
    // tile (written before the exchange barrier, read after it, before the tile is used for output)
    static constexpr int TOTAL_DOUBLES = WP_DOUBLES + TM_DOUBLES + 2 * STAGE_DOUBLES;
};

struct NoHook { __device__ __forceinline__ void operator()() const {} };

// Input accessors: local (role-oriented) segment times T(j), j = 0..HS-1, and waypoints P(j, axis),
// j = 0..HS.  The bottom role walks its half of the trajectory backwards.
template <int HS, bool BOTTOM, int TM_STRIDE = FixedLds<HS>::TM_ROW> struct LdsInputs {   // read on demand from the workgroup's LDS image
    const double *l_wp, *l_tm;
    int lane;
    __device__ __forceinline__ double T(int j) const { return l_tm[lane * TM_STRIDE + (BOTTOM ? 2 * HS - 1 - j : j)]; }
    __device__ __forceinline__ double P(int j, int ax) const { return l_wp[lane * FixedLds<HS>::WP_ROW + (BOTTOM ? 2 * HS - j : j) * 3 + ax]; }
};

// STASH: the forward sweep keeps the times/waypoints it reads in registers for the backward sweep,
// so the LDS input image is dead after the exchange barrier; `after_exchange()` runs right after
// that barrier (the persistent kernel issues the next slice's LDS-DMA there).
template <int HS, bool BOTTOM, bool STATUS, bool FULL, bool SEGMAJ, bool STASH, class In, class Hook>
__device__ __forceinline__ void fixed_body(const GenericArgs &a, int64_t b0, int64_t b, int lane,
                                           const In &in, double *stage, double *partner_stage,
                                           const Hook &after_exchange) {
    constexpr int S = 2 * HS;
    using L = FixedLds<HS>;
    const double *bc = (const double *)a.bc + (a.bc_per_traj ? b * 12 : 0);
    auto Tl = [&](int j) { return in.T(j); };
    auto Pl = [&](int j, int ax) { return in.P(j, ax); };
    const double vw = a.vw_per ? a.vw_per[b] : a.vel_zero_weight;

    // boundary derivatives (minimum_snap.cpp:527-555): vel, acc given, jerk pinned to 0;
    // time reversal negates odd derivatives
    double z[3][3], W[3][3];
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
        z[0][ax] = BOTTOM ? -bc[1 * 3 + ax] : bc[0 * 3 + ax];
        z[1][ax] = BOTTOM ? bc[3 * 3 + ax] : bc[2 * 3 + ax];
        z[2][ax] = 0.0;
    }
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) W[r][c] = 0.0;
    double Wst[HS][3][3], zst[HS][3][3];  // slot k = local waypoint k (slot 0 unused)
    bool spd = true;

    // ---- forward elimination over local interior waypoints 1..HS-1 ----
    Seg4 left, right;
    double Tst[HS], Pst[HS + 1][3];  // STASH only
    { const double t0 = Tl(0); if (STASH) Tst[0] = t0; seg4(t0, vw, left); }
    double Pa[3], Pb[3], Pc[3];  // local waypoints k-1, k, k+1
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
        Pa[ax] = Pl(0, ax);
        Pb[ax] = Pl(1, ax);
        if (STASH) { Pst[0][ax] = Pa[ax]; Pst[1][ax] = Pb[ax]; }
    }
#pragma unroll
    for (int k = 1; k < HS; ++k) {
        { const double tk = Tl(k); if (STASH) Tst[k] = tk; seg4(tk, vw, right); }
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) { Pc[ax] = Pl(k + 1, ax); if (STASH) Pst[k + 1][ax] = Pc[ax]; }
        double Sm[3][3], y[3][3], I[3][3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
#pragma unroll
            for (int c = 0; c <= r; ++c) {
                double v = ee_of(left, r, c) + right.ss[r][c];
#pragma unroll
                for (int j = 0; j < 3; ++j) v = __builtin_fma(-left.se[j][r], W[j][c], v);
                Sm[r][c] = v;
            }
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) {
                double v = left.ep[r] * (Pb[ax] - Pa[ax]);
                v = __builtin_fma(right.sp[r], Pc[ax] - Pb[ax], v);
#pragma unroll
                for (int j = 0; j < 3; ++j) v = __builtin_fma(-left.se[j][r], z[j][ax], v);
                y[r][ax] = v;
            }
        }
        spd &= inv3(Sm, I);
#pragma unroll
        for (int r = 0; r < 3; ++r) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                double w = I[r][0] * right.se[0][c];
                w = __builtin_fma(I[r][1], right.se[1][c], w);
                w = __builtin_fma(I[r][2], right.se[2][c], w);
                W[r][c] = w;
                Wst[k][r][c] = w;
            }
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) {
                double v = I[r][0] * y[0][ax];
                v = __builtin_fma(I[r][1], y[1][ax], v);
                v = __builtin_fma(I[r][2], y[2][ax], v);
                z[r][ax] = v;
                zst[k][r][ax] = v;
            }
        }
        left = right;
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) { Pa[ax] = Pb[ax]; Pb[ax] = Pc[ax]; }
    }

    CSP_STAMP(2);
    // ---- Schur carry of this half onto the middle waypoint, exchanged through LDS ----
    double Cm[3][3], cm[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int c = 0; c <= r; ++c) {
            double v = ee_of(left, r, c);
#pragma unroll
            for (int j = 0; j < 3; ++j) v = __builtin_fma(-left.se[j][r], W[j][c], v);
            Cm[r][c] = v;
        }
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
            double v = left.ep[r] * (Pb[ax] - Pa[ax]);
#pragma unroll
            for (int j = 0; j < 3; ++j) v = __builtin_fma(-left.se[j][r], z[j][ax], v);
            cm[r][ax] = v;
        }
    }
    {
        double *mine = partner_stage;  // the partner reads it from ITS tile after the barrier
        int e = 0;
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c <= r; ++c) mine[(e++) * 64 + lane] = Cm[r][c];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) mine[(e++) * 64 + lane] = cm[r][ax];
    }
    lds_barrier();
    after_exchange();
    CSP_STAMP(3);
    double xm[3][3];
    {
        const double *other = stage;
        double Sm[3][3], I[3][3];
        int e = 0;
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c <= r; ++c) {
                const double o = other[(e++) * 64 + lane];
                Sm[r][c] = Cm[r][c] + (((r + c) & 1) ? -o : o);
            }
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) {
                const double o = other[(e++) * 64 + lane];
                cm[r][ax] += (r & 1) ? o : -o;  // derivative r+1 is odd for r = 0, 2
            }
        spd &= inv3(Sm, I);
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) {
                double v = I[r][0] * cm[0][ax];
                v = __builtin_fma(I[r][1], cm[1][ax], v);
                v = __builtin_fma(I[r][2], cm[2][ax], v);
                xm[r][ax] = v;
            }
    }

    // ---- back-substitution fused with coefficient recovery, local segments HS-1 .. 0 ----
    double nanacc = 0.0;
    // bytes between consecutive trajectories' records of one segment: the default layout is
    // [B][S][3][8] (3072-byte stride at S=16); CSP_FLAG_SEGMENT_MAJOR selects [S][B][3][8]
    constexpr int RS = SEGMAJ ? 192 : S * 192;
    constexpr int ROW = L::STAGE_ROW;
    // Output leaves through a lane-major LDS tile and is read back transposed.  In the default
    // layout the records of segments (2q, 2q+1) of one trajectory form one 384-byte, 128-byte-aligned
    // run, so a wave pairs them: of the first record it stores the 128 bytes that complete a cache
    // line and holds the other 64 in the tile; with the second record it stores a 256-byte run.
    // Every line is then written whole (the single-record scheme left 1/3 of the lines half-written
    // between two bursts and measured +8 % WRITE_SIZE).  Lane maps for the three burst shapes:
    constexpr bool PAIRING = FULL && !SEGMAJ;
    const int grp = (lane * 43691) >> 19;          // 12 lanes per 192-byte record (lanes 60..63 idle)
    const int lane_in = lane - grp * 12;
    const int lds_off = grp * ROW + lane_in * 2;   // doubles
    const unsigned g_off = (unsigned)(grp * RS + lane_in * 16);  // bytes
    const int l8 = (lane >> 3) * ROW + 8 + (lane & 7) * 2;       // 8 lanes per 128-byte half, tile doubles 8..23
    const unsigned o8 = (unsigned)((lane >> 3) * RS + (lane & 7) * 16);
    const int l16 = (lane >> 4) * ROW + (lane & 15) * 2;         // 16 lanes per 256-byte run, tile doubles 0..31
    const unsigned o16 = (unsigned)((lane >> 4) * RS + (lane & 15) * 16);
    double xn[3][3];  // free derivatives at local waypoint j+1
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) xn[r][ax] = xm[r][ax];
#pragma unroll
    for (int j = HS - 1; j >= 0; --j) {
        double xk[3][3];  // free derivatives at local waypoint j
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) {
                if (j == 0) {
                    xk[r][ax] = (r == 0) ? (BOTTOM ? -bc[1 * 3 + ax] : bc[0 * 3 + ax])
                              : (r == 1) ? (BOTTOM ? bc[3 * 3 + ax] : bc[2 * 3 + ax]) : 0.0;
                } else {
                    double v = zst[j][r][ax];
                    v = __builtin_fma(-Wst[j][r][0], xn[0][ax], v);
                    v = __builtin_fma(-Wst[j][r][1], xn[1][ax], v);
                    v = __builtin_fma(-Wst[j][r][2], xn[2][ax], v);
                    xk[r][ax] = v;
                }
            }
        const double Tj = STASH ? Tst[j] : Tl(j);
        double ip[M8], tp[3];
        ip[0] = 1.0;
        ip[1] = fast_rcp(Tj);
#pragma unroll
        for (int e = 2; e < M8; ++e) ip[e] = ip[e - 1] * ip[1];
        tp[0] = Tj;
        tp[1] = Tj * Tj;
        tp[2] = tp[1] * Tj;
        const int g = BOTTOM ? S - 1 - j : j;  // global segment index
        // the middle pair of an odd half is split between the two waves: those records go out singly
        const bool paired = PAIRING && !((HS & 1) && g == (BOTTOM ? HS : HS - 1));
        const bool first = BOTTOM ? (g & 1) == 0 : (g & 1) == 1;  // first record of its pair to reach this wave
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
            double xs[3], xe[3], c[M8];
            // global orientation: the bottom role's local start is the global END, and odd
            // derivatives change sign back
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const double sgn = (BOTTOM && !(r & 1)) ? -1.0 : 1.0;
                xs[r] = BOTTOM ? sgn * xn[r][ax] : xk[r][ax];
                xe[r] = BOTTOM ? sgn * xk[r][ax] : xn[r][ax];
            }
            const double Plo = STASH ? Pst[j][ax] : Pl(j, ax), Phi = STASH ? Pst[j + 1][ax] : Pl(j + 1, ax);
            const double Ps = BOTTOM ? Phi : Plo;
            const double Pe = BOTTOM ? Plo : Phi;
            recover4(Ps, Pe - Ps, xs, xe, tp, ip, c);
            // lane-major staging tile (row = lane, 272-byte rows keep ds_write_b128 conflict-free);
            // where the axis block lands depends on the record's place in its pair (see below)
            const int tpos = !paired ? ax * M8
                           : (!BOTTOM ? (first ? (ax == 0 ? 24 : ax * M8) : ax * M8)
                                      : (first ? (ax == 2 ? 0 : 8 + ax * M8) : 8 + ax * M8));
#pragma unroll
            for (int i = 0; i < M8; i += 2) {
                double2 v2;
                v2.x = c[i];
                v2.y = c[i + 1];
                *reinterpret_cast<double2 *>(stage + lane * ROW + tpos + i) = v2;
            }
            if (STATUS) {
#pragma unroll
                for (int i = 0; i < M8; ++i) nanacc = __builtin_fma(c[i], 0.0, nanacc);
            }
        }
        // LDS operations of one wave execute in order, so the tile needs no barrier; the fences only
        // stop the compiler from reordering the (may-alias) LDS accesses.
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        if (paired) {
            // pair base = record of the even segment; TOP meets the odd record first, BOTTOM the even one
            char *pbase = reinterpret_cast<char *>((double *)a.coeffs + (b0 * S + (g & ~1)) * 24);  // uniform
            if (first) {
                double2 v[8];   // 8 rows x 128 bytes per store
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const double2 *>(stage + l8 + i * 8 * ROW);
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    *reinterpret_cast<double2 *>(pbase + (BOTTOM ? 0 : 256) + (size_t)i * 8 * RS + o8) = v[i];
            } else {
#pragma unroll
                for (int h = 0; h < 2; ++h) {  // 4 rows x 256 bytes per store, two batches of 8
                    double2 v[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const double2 *>(stage + l16 + (h * 8 + i) * 4 * ROW);
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        *reinterpret_cast<double2 *>(pbase + (BOTTOM ? 128 : 0) + (size_t)(h * 8 + i) * 4 * RS + o16) = v[i];
                }
            }
        } else {
            char *gbase = reinterpret_cast<char *>((double *)a.coeffs + (SEGMAJ ? ((int64_t)g * a.Btotal + a.Boffset + b0) : (b0 * S + g)) * 24);  // uniform
            if (FULL) {
                // 12 stores by lanes 0..59 (rows 0..59), a 13th by lanes 0..47 (rows 60..63); idle
                // lanes are masked off, not made to repeat a neighbour's piece (duplicates are traffic)
                if (lane < 60) {
                    double2 v[12];
#pragma unroll
                    for (int i = 0; i < 12; ++i)
                        v[i] = *reinterpret_cast<const double2 *>(stage + lds_off + i * 5 * ROW);
#pragma unroll
                    for (int i = 0; i < 12; ++i)
                        *reinterpret_cast<double2 *>(gbase + (size_t)i * 5 * RS + g_off) = v[i];
                    if (lane < 48)
                        *reinterpret_cast<double2 *>(gbase + (size_t)12 * 5 * RS + g_off) =
                            *reinterpret_cast<const double2 *>(stage + lds_off + 12 * 5 * ROW);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 13; ++i) {
                    const int row = i * 5 + grp;
                    if (lane < 60 && row < 64 && b0 + row < a.B) {
                        const double2 v2 = *reinterpret_cast<const double2 *>(stage + lds_off + i * 5 * ROW);
                        *reinterpret_cast<double2 *>(gbase + (size_t)i * 5 * RS + g_off) = v2;
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) xn[r][ax] = xk[r][ax];
    }
    CSP_STAMP(4);
    CSP_STAMP_RT(6);
    if (STATUS && b0 + lane < a.B) {
        const int bits = (spd ? 0 : 2) | ((nanacc == 0.0) ? 0 : 1);
        if (bits) atomicOr(a.status + b, bits);
    }
}

// FULL = every workgroup owns 64 real trajectories (B % 64 == 0); the ragged remainder of a
// batch is a second, single-workgroup launch of the FULL=false variant.
template <int HS, bool STATUS, bool FULL, bool SEGMAJ>
__global__ void __launch_bounds__(128) minsnap_fixed_kernel(GenericArgs a) {
    using L = FixedLds<HS>;
    constexpr int S = 2 * HS;
    __shared__ __attribute__((aligned(16))) double lds[L::TOTAL_DOUBLES];
    double *l_wp = lds;
    double *l_tm = l_wp + L::WP_DOUBLES;
    double *l_stage = l_tm + L::TM_DOUBLES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int role = tid >> 6;  // wave-uniform
    const int64_t b0 = (int64_t)blockIdx.x * 64;
    const int rows = (int)((a.B - b0) < 64 ? (a.B - b0) : 64);

    CSP_STAMP_RT(5);
    CSP_STAMP(0);
    // ---- coalesced copy-in: the workgroup's waypoints and times are contiguous in HBM ----
    {
        const double2 *g_wp = reinterpret_cast<const double2 *>((const double *)a.wp + b0 * L::WP_ROW);
        const int n_wp = rows * L::WP_ROW / 2;  // 16-byte pieces (WP_ROW*64 is even; a ragged tail row count keeps it even too)
        constexpr int WP_ITERS = (64 * L::WP_ROW / 2 + 127) / 128;
#pragma unroll
        for (int it = 0; it < WP_ITERS; ++it) {
            const int c = it * 128 + tid;
            if (c < n_wp) reinterpret_cast<double2 *>(l_wp)[c] = g_wp[c];
        }
        if ((rows * L::WP_ROW) & 1) {  // odd number of doubles: last one by itself
            if (tid == 0) l_wp[rows * L::WP_ROW - 1] = ((const double *)a.wp + b0 * L::WP_ROW)[rows * L::WP_ROW - 1];
        }
        const double2 *g_tm = reinterpret_cast<const double2 *>((const double *)a.times + b0 * S);
        const int n_tm = rows * HS;  // 16-byte pieces
        constexpr int TM_ITERS = (64 * HS + 127) / 128;
#pragma unroll
        for (int it = 0; it < TM_ITERS; ++it) {
            const int c = it * 128 + tid;
            const int row = c / HS, col = c - row * HS;
            if (c < n_tm) *reinterpret_cast<double2 *>(l_tm + row * L::TM_ROW + col * 2) = g_tm[c];
        }
    }
    __syncthreads();
    CSP_STAMP(1);

    int64_t b = b0 + lane;
    if (b >= a.B) b = a.B - 1;  // idle lanes of a ragged last workgroup: harmless, store nothing
    if (role == 0) {
        const LdsInputs<HS, false> in{l_wp, l_tm, lane};
        fixed_body<HS, false, STATUS, FULL, SEGMAJ, false>(a, b0, b, lane, in, l_stage, l_stage + L::STAGE_DOUBLES, NoHook{});
    } else {
        const LdsInputs<HS, true> in{l_wp, l_tm, lane};
        fixed_body<HS, true, STATUS, FULL, SEGMAJ, false>(a, b0, b, lane, in, l_stage + L::STAGE_DOUBLES, l_stage, NoHook{});
    }
}

// LDS-DMA (global_load_lds_dwordx4: HBM -> LDS, no registers in between) of one 64-trajectory
// slice: waypoints then times, copied linearly in 16-byte pieces, 1 KiB per wave instruction.
template <int HS> struct SlicePrefetch {
    typedef const __attribute__((address_space(1))) void *gptr_t;
    typedef __attribute__((address_space(3))) void *lptr_t;
    typedef __attribute__((address_space(3))) char *lchar_t;
    using L = FixedLds<HS>;
    static constexpr int S = 2 * HS;
    static constexpr int WP_PIECES = 64 * L::WP_ROW / 2;   // 64*WP_ROW is even
    static constexpr int TM_PIECES = 64 * S / 2;
    static constexpr int WP_ITERS = (WP_PIECES + 127) / 128, TM_ITERS = (TM_PIECES + 127) / 128;
    static constexpr int TM_BYTE_OFF = 64 * L::WP_ROW * 8;
    const char *wp, *tm;     // batch base pointers
    lchar_t lds3;            // LDS image base (waypoints, then unpadded times)
    int tid, role;
    int64_t next, n_slices;
    __device__ __forceinline__ void issue(int64_t slice) const {
        const char *g_wp = wp + slice * (64 * L::WP_ROW * 8);
        const char *g_tm = tm + slice * (64 * S * 8);
#pragma unroll
        for (int it = 0; it < WP_ITERS; ++it) {
            const int q = it * 128 + tid;  // piece index; a wave's 64 pieces are contiguous
            if (q < WP_PIECES)
                __builtin_amdgcn_global_load_lds((gptr_t)(g_wp + (size_t)q * 16), (lptr_t)(lds3 + (it * 128 + role * 64) * 16), 16, 0, 0);
        }
#pragma unroll
        for (int it = 0; it < TM_ITERS; ++it) {
            const int q = it * 128 + tid;
            if (q < TM_PIECES)
                __builtin_amdgcn_global_load_lds((gptr_t)(g_tm + (size_t)q * 16), (lptr_t)(lds3 + TM_BYTE_OFF + (it * 128 + role * 64) * 16), 16, 0, 0);
        }
    }
    __device__ __forceinline__ void operator()() const { if (next < n_slices) issue(next); }
};

template <int HS, bool BOTTOM, bool STATUS, bool SEGMAJ>
__device__ __forceinline__ void persistent_role_loop(const GenericArgs &a, int n_slices, int lane, const double *l_wp,
                                                     const double *l_tm, double *stage, double *partner_stage,
                                                     SlicePrefetch<HS> pf) {
    constexpr int S = 2 * HS;
    // Vector-memory operations a wave issues AFTER a slice's prefetch and before the next top-of-loop
    // wait: its store bursts.  Must not be over-estimated (the counted wait below relies on at least
    // this many younger operations existing).  Paired records: 8 + 16 stores per pair; single
    // records (segment-major layout, or the straddling middle pair of an odd half): 13 each.
    constexpr int PAIRS = SEGMAJ ? 0 : HS / 2;
    constexpr int STORES_PER_SLICE = PAIRS * 24 + (HS - 2 * PAIRS) * 13;
    bool first = true;
    for (int64_t slice = blockIdx.x; slice < n_slices; slice += gridDim.x) {
        // the prefetch of this slice is older than every store of the previous slice, so waiting for
        // all but the youngest min(63, stores) operations covers it without draining the stores
        if (first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(STORES_PER_SLICE < 63 ? STORES_PER_SLICE : 63) : "memory");
        lds_barrier();
        if (first) CSP_STAMP(1);
        const LdsInputs<HS, BOTTOM, S> in{l_wp, l_tm, lane};  // unpadded rows: LDS-DMA writes linearly
        const int64_t b0 = slice * 64;
        pf.next = slice + gridDim.x;
        // the image is dead once both waves passed the exchange barrier: prefetch the next slice there
        fixed_body<HS, BOTTOM, STATUS, true, SEGMAJ, true>(a, b0, b0 + lane, lane, in, stage, partner_stage, pf);
        first = false;
    }
}

// Persistent variant for the full workgroups of a batch: gridDim.x workgroups (two per CU) walk the
// batch with stride gridDim.x; the NEXT slice's inputs stream into LDS while the current slice is
// back-substituted and stored, so only a workgroup's very first copy-in is exposed.
template <int HS, bool STATUS, bool SEGMAJ>
__global__ void __launch_bounds__(128) minsnap_fixed_persistent_kernel(GenericArgs a, int n_slices) {
    using L = FixedLds<HS>;
    constexpr int S = 2 * HS;
    __shared__ __attribute__((aligned(16))) double lds[64 * L::WP_ROW + 64 * S + 2 * L::STAGE_DOUBLES];
    double *l_wp = lds;
    double *l_tm = l_wp + 64 * L::WP_ROW;
    double *l_stage = l_tm + 64 * S;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int role = tid >> 6;  // wave-uniform
    SlicePrefetch<HS> pf;
    pf.wp = reinterpret_cast<const char *>(a.wp);
    pf.tm = reinterpret_cast<const char *>(a.times);
    pf.lds3 = (typename SlicePrefetch<HS>::lchar_t)lds;  // cast straight from the LDS object
    pf.tid = tid;
    pf.role = role;
    pf.next = 0;
    pf.n_slices = n_slices;
    CSP_STAMP_RT(5);
    CSP_STAMP(0);
    if ((int64_t)blockIdx.x < n_slices) pf.issue(blockIdx.x);
    // one loop per role: each wave's instruction stream holds a single specialisation
    if (role == 0) persistent_role_loop<HS, false, STATUS, SEGMAJ>(a, n_slices, lane, l_wp, l_tm, l_stage, l_stage + L::STAGE_DOUBLES, pf);
    else persistent_role_loop<HS, true, STATUS, SEGMAJ>(a, n_slices, lane, l_wp, l_tm, l_stage + L::STAGE_DOUBLES, l_stage, pf);
}

}  // namespace

bool fixed_supported(int order, int S, bool f32, double path_weight, bool ragged) {
    return order == 4 && !f32 && !ragged && path_weight == 0.0 && S >= 2 && S <= 16 && (S % 2) == 0;
}

const char *fixed_kernel_name(int S) {
    switch (S) {
        case 2: return "fixed_o4_s2_f64";
        case 4: return "fixed_o4_s4_f64";
        case 6: return "fixed_o4_s6_f64";
        case 8: return "fixed_o4_s8_f64";
        case 10: return "fixed_o4_s10_f64";
        case 12: return "fixed_o4_s12_f64";
        case 14: return "fixed_o4_s14_f64";
        case 16: return "fixed_o4_s16_f64";
    }
    return "fixed_unavailable";
}

hipError_t launch_fixed(const GenericArgs &a, hipStream_t st) {
    if (a.B == 0) return hipSuccess;
    hipError_t e;
    if (a.status && (e = hipMemsetAsync(a.status, 0, sizeof(int32_t) * (size_t)a.B, st)) != hipSuccess) return e;
    // without the path penalty the reference's deviation metric is evaluated at t* = 0, where the
    // polynomial equals its waypoint exactly (minimum_snap.cpp:342, :596-617)
    if (a.max_dev && (e = hipMemsetAsync(a.max_dev, 0, sizeof(double) * (size_t)a.B, st)) != hipSuccess) return e;
    const int64_t n_full = a.B / 64, rem = a.B % 64;
    const dim3 block(128);
    // persistent grid: two workgroups per CU (register- and LDS-limited residency of this kernel)
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    const int64_t pgrid = n_full < 2 * (int64_t)cus ? n_full : 2 * (int64_t)cus;
    GenericArgs t = a;  // tail: the last B % 64 trajectories, one workgroup
    if (rem) {
        const int64_t off = n_full * 64;
        t.B = rem;
        t.wp = (const double *)a.wp + off * (a.S + 1) * 3;
        t.times = (const double *)a.times + off * a.S;
        if (a.seg_major) t.Boffset = off;   // segment-major records are addressed from the batch start
        else t.coeffs = (double *)a.coeffs + off * a.S * 24;
        if (a.bc_per_traj) t.bc = (const double *)a.bc + off * 12;
        if (a.status) t.status = a.status + off;
        if (a.vw_per) t.vw_per = a.vw_per + off;
    }
#define CSP_FIXED_PERSIST(hs, st_, args_)                                                                       \
    do {                                                                                                        \
        if (a.seg_major)                                                                                        \
            hipLaunchKernelGGL((minsnap_fixed_persistent_kernel<hs, st_, true>), dim3((unsigned)pgrid), block, 0, st, args_, (int)n_full);  \
        else                                                                                                    \
            hipLaunchKernelGGL((minsnap_fixed_persistent_kernel<hs, st_, false>), dim3((unsigned)pgrid), block, 0, st, args_, (int)n_full); \
    } while (0)
#define CSP_FIXED_LAUNCH(hs, st_, full_, grid_, args_)                                                          \
    do {                                                                                                        \
        if (a.seg_major)                                                                                        \
            hipLaunchKernelGGL((minsnap_fixed_kernel<hs, st_, full_, true>), dim3((unsigned)(grid_)), block, 0, st, args_);  \
        else                                                                                                    \
            hipLaunchKernelGGL((minsnap_fixed_kernel<hs, st_, full_, false>), dim3((unsigned)(grid_)), block, 0, st, args_); \
    } while (0)
#define CSP_FIXED_CASE(hs)                                                              \
    case 2 * hs:                                                                        \
        if (n_full) {                                                                   \
            GenericArgs f = a;                                                          \
            f.B = n_full * 64;                                                          \
            if (a.persistent) {                                                         \
                if (a.status) CSP_FIXED_PERSIST(hs, true, f);                           \
                else CSP_FIXED_PERSIST(hs, false, f);                                   \
            } else if (a.status) CSP_FIXED_LAUNCH(hs, true, true, n_full, f);           \
            else CSP_FIXED_LAUNCH(hs, false, true, n_full, f);                          \
        }                                                                               \
        if (rem) {                                                                      \
            if (a.status) CSP_FIXED_LAUNCH(hs, true, false, 1, t);                      \
            else CSP_FIXED_LAUNCH(hs, false, false, 1, t);                              \
        }                                                                               \
        break;
    switch (a.S) {
        CSP_FIXED_CASE(1) CSP_FIXED_CASE(2) CSP_FIXED_CASE(3) CSP_FIXED_CASE(4)
        CSP_FIXED_CASE(5) CSP_FIXED_CASE(6) CSP_FIXED_CASE(7) CSP_FIXED_CASE(8)
        default: return hipErrorInvalidValue;
    }
#undef CSP_FIXED_CASE
#undef CSP_FIXED_LAUNCH
#undef CSP_FIXED_PERSIST
    return hipGetLastError();
}

}  // namespace csp
