// minsnap_span.hip -- workspace-free kernel for LONG trajectories (16 < S <= 1024, orders 2..5, fp64 or
// fp32 storage with fp64 arithmetic, zero-velocity penalty, no path penalty).
//
// Same substructuring as minsnap_chunked.hip (chunk Schur complements -> interface system in LDS ->
// chunk solve), but a lane owns a SPAN of up to 16 segments, so a 64-segment trajectory has 3
// interfaces instead of 15 and the redundant interface solve -- two thirds of the chunked kernel's
// instructions at that length -- all but disappears.  A span's 15 interior waypoints do not fit in
// registers as factors (W_k, z_k: 18 doubles each at order 4), so nothing about them is kept:
//   1. Schur complement of the span: two carry-only eliminations over its interior waypoints,
//      left-to-right with the left interface as a parameter and right-to-left on the time-reversed
//      span (run-time loops; times and waypoints come from L1/L2, one step ahead of their use);
//   2. interface system in LDS, twisted elimination (identical to the chunked kernel);
//   3. recovery by RECOMPUTATION: the forward elimination from the (now known) left end is repeated,
//      each time stopping three waypoints earlier and keeping the factors of its last three waypoints
//      only; those are back-substituted from the already known waypoint on their right and their
//      segments are recovered and stored.  15 + 12 + 9 + 6 + 3 = 45 elimination steps instead of 15,
//      against zero bytes of workspace and 54 doubles of registers.
// Per 16 segments a lane issues ~13.6 k VALU instructions: 0.85 k per trajectory-segment-wave against
// the chunked kernel's 1.8 k (measured: DESIGN.md section 5.2c).
#include "minsnap_iface.h"

namespace csp {
namespace span {

using fixedk::Seg;
using fixedk::SmallSpd;
using fixedk::ee_of;
using fixedk::seg_make;

constexpr int CSEG = 16;  // segments per lane

template <typename IO> __device__ __forceinline__ double ld(const IO *p) { return (double)*p; }

// This lane's span seen in local order (REV: time-reversed): segment times T(i), waypoints P(i, ax).
template <typename IO> struct SpanView {
    const IO *wp, *tm;
    int64_t pt0, sg0;
    int c;
    bool rev;
    __device__ __forceinline__ double T(int i) const { return ld(tm + sg0 + (rev ? c - 1 - i : i)); }
    __device__ __forceinline__ void P(int i, double (&p)[3]) const {
        const IO *q = wp + (pt0 + (rev ? c - i : i)) * 3;
        p[0] = ld(q); p[1] = ld(q + 1); p[2] = ld(q + 2);
    }
};

// One elimination step at local waypoint k between `left` (segment k-1) and `right` (segment k):
// [W | z | V] <- S^-1 [se_right | y - se_left^T z | -se_left^T V],  S = ee_left + ss_right - se_left^T W.
template <int O, bool CROSS>
__device__ __forceinline__ bool elim_step(const Seg<O> &left, const Seg<O> &right, const double (&Pa)[3], const double (&Pb)[3],
                                          const double (&Pc)[3], double (&W)[O - 1][O - 1], double (&z)[O - 1][3],
                                          double (&V)[O - 1][O - 1]) {
    constexpr int N = O - 1, NC = N + 3 + (CROSS ? N : 0);
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
            v = __builtin_fma(right.sp[r], Pc[ax] - Pb[ax], v);
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
                R[r][N + 3 + q] = v;
            }
        }
    }
    const bool ok = SmallSpd<N, NC>::solve(Sm, R);
#pragma unroll
    for (int r = 0; r < N; ++r) {
#pragma unroll
        for (int q = 0; q < N; ++q) { W[r][q] = R[r][q]; if (CROSS) V[r][q] = R[r][N + 3 + q]; }
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) z[r][ax] = R[r][N + ax];
    }
    return ok;
}

// Step 1: eliminates the span's interior waypoints in local order with the START interface x_s as a
// parameter, carrying only the current factor; returns the span's part of the END interface row
//   D x_e + X x_s = r      (X only when CROSS; it is E^T for the forward direction).
template <int O, bool CROSS, typename IO>
__device__ __forceinline__ bool span_schur(const SpanView<IO> &v, double vw, double (&D)[O - 1][O - 1], double (&rr)[O - 1][3],
                                           double (&X)[O - 1][O - 1]) {
    constexpr int N = O - 1;
    double W[N][N], z[N][3], V[N][N];
#pragma unroll
    for (int r = 0; r < N; ++r) {
#pragma unroll
        for (int q = 0; q < N; ++q) { W[r][q] = 0.0; V[r][q] = (r == q) ? -1.0 : 0.0; }
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) z[r][ax] = 0.0;
    }
    const int c = v.c;
    Seg<O> left, right;
    seg_make<O>(c > 0 ? v.T(0) : 1.0, vw, left);
    double Pa[3] = {0, 0, 0}, Pb[3] = {0, 0, 0}, Pc[3] = {0, 0, 0};
    if (c > 0) { v.P(0, Pa); v.P(1, Pb); }
    double Tn = c > 1 ? v.T(1) : 1.0;   // one step ahead of their use
    if (c > 1) v.P(2, Pc);
    bool spd = true;
    for (int k = 1; k < c; ++k) {
        const double Tk = Tn;
        double Pk[3] = {Pc[0], Pc[1], Pc[2]};
        // the NEXT step's inputs, requested now and needed one iteration later.  Unconditional (index clamped to the last
        // step) on purpose: inside `if (k + 1 < c)` hipcc waited for the loads right there (s_waitcnt vmcnt(0) at the end of
        // the conditional block), which put one memory round trip on every elimination step
        { const int kn = k + 1 < c ? k + 1 : c - 1; Tn = v.T(kn); v.P(kn + 1, Pc); }
        seg_make<O>(Tk, vw, right);
        spd &= elim_step<O, CROSS>(left, right, Pa, Pb, Pk, W, z, V);
        left = right;
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) { Pa[ax] = Pb[ax]; Pb[ax] = Pk[ax]; }
    }
#pragma unroll
    for (int r = 0; r < N; ++r) {
#pragma unroll
        for (int q = 0; q < N; ++q) {
            double vv = ee_of<O>(left, r, q), x = 0.0;
#pragma unroll
            for (int j = 0; j < N; ++j) {
                vv = __builtin_fma(-left.se[j][r], W[j][q], vv);
                if (CROSS) x = __builtin_fma(-left.se[j][r], V[j][q], x);
            }
            D[r][q] = vv;
            if (CROSS) X[r][q] = x;
        }
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
            double vv = left.ep[r] * (Pb[ax] - Pa[ax]);
#pragma unroll
            for (int j = 0; j < N; ++j) vv = __builtin_fma(-left.se[j][r], z[j][ax], vv);
            rr[r][ax] = vv;
        }
    }
    return spd;
}

using iface::IfaceLds;
using iface::store_axis;

// STATUS = the caller passed a `status` array: only then are the coefficients tested for NaN/Inf (on the values
// as stored, so that an fp32 overflow is caught: two conversions and an fma per coefficient otherwise spent for nothing).
template <int O, typename IO, bool STATUS>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(O <= 4 ? 2 : 1)))
minsnap_span_kernel(GenericArgs a, int lpt_log2) {
    constexpr int N = O - 1, M = 2 * O;
    using IL = IfaceLds<O>;
    __shared__ __attribute__((aligned(16))) double lds[IL::STAGE_DOUBLES];   // interface image, later the store staging rows
    __shared__ double xch[3 * N * 64];   // solved interface derivatives, handed to the right-hand neighbour
    __shared__ unsigned long long l_addr[64];   // where each lane's parked record goes
    const int lane = threadIdx.x;
    const int lpt = 1 << lpt_log2;
    const int j = lane & (lpt - 1);                                  // my span
    const int64_t b = ((int64_t)blockIdx.x * 64 + lane) >> lpt_log2;  // my trajectory
    const bool traj_ok = b < a.B;
    const int64_t bb = traj_ok ? b : a.B - 1;
    int64_t seg0;
    int S;
    if (a.seg_off) { seg0 = a.seg_off[bb]; S = (int)(a.seg_off[bb + 1] - seg0); }
    else { seg0 = bb * (int64_t)a.S; S = a.S; }
    if (!traj_ok) S = 0;
    // spans in use: as few as 16 segments each allow, balanced
    int nch = (S + CSEG - 1) / CSEG;
    if (nch > lpt) nch = lpt;
    const int q = nch > 0 ? S / nch : 0, rem = nch > 0 ? S - q * nch : 0;
    const bool active = j < nch;
    const int c = active ? q + (j < rem ? 1 : 0) : 0;            // my segments: s0 .. s0+c-1 (<= 16)
    const int s0 = j * q + (j < rem ? j : rem);
    const IO *wp = (const IO *)a.wp, *tm = (const IO *)a.times;
    const int64_t pt0 = seg0 + bb + s0, sg0 = seg0 + s0;
    const double vw = a.vw_per ? a.vw_per[bb] : a.vel_zero_weight;
    bool spd = true;

    // ---- step 0: pull the span's inputs into the caches with independent loads ----
    // The sweeps below read a waypoint and a time per elimination step, inside the dependency chain;
    // their first touch would pay one HBM latency per cache line, serially.  Touch every line once,
    // all loads in flight together (one waypoint coordinate each: waypoints are 24 B apart).
    {
        // (ONE branch for the lane, indices clamped into the span: a conditional per load would make hipcc wait for each
        // of them in turn -- the opposite of what this step is for)
        IO tv[CSEG + 1], tt[CSEG / 4];
        double sink = 0.0;
        if (active && c > 0) {
#pragma unroll
            for (int i = 0; i <= CSEG; ++i) tv[i] = wp[(pt0 + (i <= c ? i : c)) * 3];
#pragma unroll
            for (int i = 0; i < CSEG / 4; ++i) tt[i] = tm[sg0 + (4 * i < c ? 4 * i : c - 1)];
#pragma unroll
            for (int i = 0; i <= CSEG; ++i) sink += (double)tv[i];
#pragma unroll
            for (int i = 0; i < CSEG / 4; ++i) sink += (double)tt[i];
        }
        if (sink == 1.0e-300) xch[lane] = sink;   // never true for real data: keeps the loads alive
    }

    // ---- step 1: the span's Schur complement onto its two interfaces ----
    double DR[N][N], rR[N][3];
    {
        double DL[N][N], rL[N][3], Et[N][N], unused[N][N];
        const SpanView<IO> rv{wp, tm, pt0, sg0, c, true};
        spd &= span_schur<O, false>(rv, vw, DL, rL, unused);            // reversed frame: the START interface row
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
        const SpanView<IO> fv{wp, tm, pt0, sg0, c, false};
        spd &= span_schur<O, true>(fv, vw, DR, rR, Et);                 // end row: DR x_R + Et x_L = rR
        if (active) {
#pragma unroll
            for (int r = 0; r < N; ++r)
#pragma unroll
                for (int qq = 0; qq < N; ++qq) lds[(IL::OFF_E + r * N + qq) * 64 + lane] = Et[qq][r];       // E = Et^T
        }
    }
    __syncthreads();
    // interface i (1 <= i <= nch-1) sums span i-1's end row and span i's start row: lane i-1 adds its part
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

    // ---- step 3: recovery by recomputation, three waypoints (and their segments) per pass ----
    // Records leave through LDS (the interface image is dead now): a lane parks the record of the segment
    // it just recovered in its staging row, then the wave stores all parked records together, PPR lanes
    // per record in 16-byte (8-byte: fp32 at odd order) pieces -- contiguous 1.5-line bursts instead of
    // 64 scattered 16-byte writes per instruction (which capped this kernel at 1.3 TB/s).
    constexpr int RECB = 3 * M * (int)sizeof(IO);            // record bytes
    constexpr int PIECE = (RECB % 16 == 0) ? 16 : 8;
    constexpr int PPR = RECB / PIECE, RPI = 64 / PPR, NI = (64 + RPI - 1) / RPI;
    constexpr int RB = RECB + 16;                            // staging row stride (bytes)
    static_assert(RB * 64 <= (int)sizeof(double) * IL::STAGE_DOUBLES, "staging rows must fit the interface image");
    char *stage = reinterpret_cast<char *>(lds);
    double nanacc = 0.0;
    const SpanView<IO> fv{wp, tm, pt0, sg0, c, false};
    IO *co = (IO *)a.coeffs + sg0 * (int64_t)(3 * M);
    auto park = [&](int g, const double (&xs)[N][3], const double (&xe)[N][3]) {   // segment g: waypoints g, g+1
        const double Ts = fv.T(g);
        double P0[3], P1[3];
        fv.P(g, P0);
        fv.P(g + 1, P1);
        double ip[M], tp[N];
        ip[0] = 1.0;
        ip[1] = fast_rcp(Ts);
#pragma unroll
        for (int e = 2; e < M; ++e) ip[e] = ip[e - 1] * ip[1];
        tp[0] = Ts;
#pragma unroll
        for (int e = 1; e < N; ++e) tp[e] = tp[e - 1] * Ts;
        IO *row = reinterpret_cast<IO *>(stage + lane * RB);
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
            double a0[N], a1[N], cc[M];
#pragma unroll
            for (int r = 0; r < N; ++r) { a0[r] = xs[r][ax]; a1[r] = xe[r][ax]; }
            fixedk::recover<O>(P0[ax], P1[ax] - P0[ax], a0, a1, tp, ip, cc);
            store_axis<IO, M>(row + ax * M, cc);
            if (STATUS) {
#pragma unroll
                for (int i = 0; i < M; ++i) nanacc = __builtin_fma((double)(IO)cc[i], 0.0, nanacc);
            }
        }
        l_addr[lane] = reinterpret_cast<unsigned long long>(co + (int64_t)g * (3 * M));
    };
    auto flush = [&](bool has) {
        __syncthreads();
        const unsigned long long pend = __builtin_amdgcn_ballot_w64(has);
        const int grp = lane / PPR, pc = lane - grp * PPR;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int sl = i * RPI + grp;   // source lane of the record this group stores
            if (lane < RPI * PPR && sl < 64 && ((pend >> sl) & 1ull)) {
                char *dst = reinterpret_cast<char *>(l_addr[sl]) + pc * PIECE;
                const char *src = stage + sl * RB + pc * PIECE;
                if (PIECE == 16) *reinterpret_cast<double2 *>(dst) = *reinterpret_cast<const double2 *>(src);
                else *reinterpret_cast<double *>(dst) = *reinterpret_cast<const double *>(src);
            }
        }
        __syncthreads();
    };
    double xnx[N][3];     // free derivatives at local waypoint `hi` (known)
#pragma unroll
    for (int r = 0; r < N; ++r)
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) xnx[r][ax] = xR[r][ax];
    int hi = c;
    while (__builtin_amdgcn_ballot_w64(active && hi > 1) != 0) {
        const bool on = active && hi > 1;
        const int lo = hi - 3;    // first waypoint whose factors are kept (may be < 1)
        double Wst[3][N][N], zst[3][N][3];
        if (on) {
            // forward elimination from the known left end up to waypoint hi-1; factors kept for hi-3..hi-1
            double W[N][N], z[N][3], unusedV[N][N];
#pragma unroll
            for (int r = 0; r < N; ++r) {
#pragma unroll
                for (int qq = 0; qq < N; ++qq) W[r][qq] = 0.0;
#pragma unroll
                for (int ax = 0; ax < 3; ++ax) z[r][ax] = xL[r][ax];
            }
            Seg<O> left, right;
            seg_make<O>(fv.T(0), vw, left);
            double Pa[3], Pb[3], Pc[3] = {0, 0, 0};
            fv.P(0, Pa);
            fv.P(1, Pb);
            double Tn = fv.T(1);      // hi > 1, so segment 1 exists
            fv.P(2, Pc);
            for (int k = 1; k < lo; ++k) {
                const double Tk = Tn;
                double Pk[3] = {Pc[0], Pc[1], Pc[2]};
                { const int kn = k + 1 < hi ? k + 1 : hi - 1; Tn = fv.T(kn); fv.P(kn + 1, Pc); }   // unconditional, see span_schur
                seg_make<O>(Tk, vw, right);
                spd &= elim_step<O, false>(left, right, Pa, Pb, Pk, W, z, unusedV);
                left = right;
#pragma unroll
                for (int ax = 0; ax < 3; ++ax) { Pa[ax] = Pb[ax]; Pb[ax] = Pk[ax]; }
            }
#pragma unroll
            for (int s3 = 0; s3 < 3; ++s3) {
                const int k = lo + s3;
                if (k >= 1) {
                    const double Tk = Tn;
                    double Pk[3] = {Pc[0], Pc[1], Pc[2]};
                    { const int kn = k + 1 < hi ? k + 1 : hi - 1; Tn = fv.T(kn); fv.P(kn + 1, Pc); }
                    seg_make<O>(Tk, vw, right);
                    spd &= elim_step<O, false>(left, right, Pa, Pb, Pk, W, z, unusedV);
                    left = right;
#pragma unroll
                    for (int ax = 0; ax < 3; ++ax) { Pa[ax] = Pb[ax]; Pb[ax] = Pk[ax]; }
#pragma unroll
                    for (int r = 0; r < N; ++r) {
#pragma unroll
                        for (int qq = 0; qq < N; ++qq) Wst[s3][r][qq] = W[r][qq];
#pragma unroll
                        for (int ax = 0; ax < 3; ++ax) zst[s3][r][ax] = z[r][ax];
                    }
                }
            }
        }
        // back-substitution of waypoints hi-1, hi-2, hi-3 and recovery of the segments on their right
#pragma unroll
        for (int s3 = 2; s3 >= 0; --s3) {
            const int k = lo + s3;
            const bool has = on && k >= 1;
            if (has) {
                double xk[N][3];
#pragma unroll
                for (int r = 0; r < N; ++r)
#pragma unroll
                    for (int ax = 0; ax < 3; ++ax) {
                        double v = zst[s3][r][ax];
#pragma unroll
                        for (int kk = 0; kk < N; ++kk) v = __builtin_fma(-Wst[s3][r][kk], xnx[kk][ax], v);
                        xk[r][ax] = v;
                    }
                park(k, xk, xnx);
#pragma unroll
                for (int r = 0; r < N; ++r)
#pragma unroll
                    for (int ax = 0; ax < 3; ++ax) xnx[r][ax] = xk[r][ax];
            }
            flush(has);
        }
        if (on) hi -= 3;
    }
    if (active) park(0, xL, xnx);   // the first segment: left end known, waypoint 1 (or the right end when c == 1) in xnx
    flush(active);
    if (STATUS && active) {
        const int bits = (spd ? 0 : 2) | ((nanacc == 0.0) ? 0 : 1);
        if (bits) atomicOr(a.status + b, bits);
    }
}

template <int O> hipError_t launch_o(const GenericArgs &a, bool f32, int lpt_log2, hipStream_t st) {
    const int64_t lanes = a.B << lpt_log2;
    const dim3 grid((unsigned)((lanes + 63) / 64)), block(64);
    if (a.status) {
        if (f32) hipLaunchKernelGGL((minsnap_span_kernel<O, float, true>), grid, block, 0, st, a, lpt_log2);
        else hipLaunchKernelGGL((minsnap_span_kernel<O, double, true>), grid, block, 0, st, a, lpt_log2);
    } else {
        if (f32) hipLaunchKernelGGL((minsnap_span_kernel<O, float, false>), grid, block, 0, st, a, lpt_log2);
        else hipLaunchKernelGGL((minsnap_span_kernel<O, double, false>), grid, block, 0, st, a, lpt_log2);
    }
    return hipGetLastError();
}

}  // namespace span

int span_lanes_log2(int Smax) {
    int l = 0;
    while ((span::CSEG << l) < Smax) ++l;
    return l;
}

bool span_supported(int order, int Smax, bool f32_arith, double path_weight, bool seg_major) {
    return order >= 2 && order <= 5 && Smax > 16 && Smax <= span::CSEG * 64 && !f32_arith && path_weight == 0.0 && !seg_major;
}

hipError_t launch_span(const GenericArgs &a, bool f32, int Smax, hipStream_t st) {
    if (a.B == 0) return hipSuccess;
    hipError_t e;
    if (a.status && (e = hipMemsetAsync(a.status, 0, sizeof(int32_t) * (size_t)a.B, st)) != hipSuccess) return e;
    // no path penalty: the deviation metric is evaluated at t* = 0 where it vanishes (minimum_snap.cpp:342)
    if (a.max_dev && (e = hipMemsetAsync(a.max_dev, 0, sizeof(double) * (size_t)a.B, st)) != hipSuccess) return e;
    const int l = span_lanes_log2(Smax);
    switch (a.order) {
        case 2: return span::launch_o<2>(a, f32, l, st);
        case 3: return span::launch_o<3>(a, f32, l, st);
        case 4: return span::launch_o<4>(a, f32, l, st);
        case 5: return span::launch_o<5>(a, f32, l, st);
    }
    return hipErrorInvalidValue;
}

}  // namespace csp
