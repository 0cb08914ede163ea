// minsnap_generic.hip -- generic (any S, order 1..5, f64/f32, ragged, penalties) kernel.
//
// One lane owns one trajectory and walks its interior waypoints: a block-tridiagonal LDL^T
// sweep (forward), then back-substitution fused with coefficient recovery.  The per-waypoint
// factors W_k = S_k^-1 C_k and z_k = S_k^-1 y_k live in a device workspace laid out
// [waypoint][entry][trajectory], so the 64 lanes of a wave touch 64 consecutive words.
// This kernel is the correctness workhorse: it carries every option of the reference's
// SolveQPClosedForm (path penalty :347-469, zero-velocity penalty :473-509, deviation metric
// :594-624).  The headline fixed-size buckets are served by minsnap_fixed.hip.
#include "minsnap_device.h"
#include "minsnap_launch.h"

namespace csp {

// IO = storage type of waypoints/times/bc/coeffs, R = arithmetic (and workspace) type.
template <typename IO, typename R>
__device__ __forceinline__ void load3(const IO *p, R (&v)[3]) { v[0] = R(p[0]); v[1] = R(p[1]); v[2] = R(p[2]); }

// One axis record (M = 2*order values) as 16-byte (f64) or 8-byte (f32) vector stores: record bases
// are multiples of M*sizeof(IO), i.e. 16*order resp. 8*order bytes.
template <int M, typename IO> __device__ __forceinline__ void store_row(IO *dst, const IO (&q)[M]) {
    typedef IO vec2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int i = 0; i < M; i += 2) {
        vec2 v;
        v.x = q[i];
        v.y = q[i + 1];
        *reinterpret_cast<vec2 *>(dst + i) = v;
    }
}

template <int O, typename IO> struct TrajView {
    const IO *wp;    // [(S+1)][3]
    const IO *tm;    // [S]
    IO *co;          // segment k's record at co + k*seg_stride: [3][2O]
    int64_t seg_stride;  // 3*2O (trajectory-major) or B*3*2O (CSP_FLAG_SEGMENT_MAJOR)
    int S;
};

// One full solve pass.  PATH selects the penalised system (second pass of :347-469).
// Returns status bits; writes coefficients; if dev_out != nullptr also the deviation metric.
// Both sweeps are software-pipelined: the inputs (and, backwards, the stored factors) of the NEXT
// waypoint are requested before the current one is processed, so one lane's chain of dependent
// global loads overlaps with its arithmetic.
template <int O, typename IO, typename R, bool PATH>
__device__ int solve_pass(const TrajView<O, IO> &tv, const R (&x0)[(O > 1 ? O - 1 : 1)][3],
                          const R (&xS)[(O > 1 ? O - 1 : 1)][3], R pw, R vw,
                          R *ws, const int *tstar, int64_t B, int64_t b, double *dev_out) {
    constexpr int N = O - 1;
    constexpr int NS = (O > 1) ? O - 1 : 1;
    constexpr int M = 2 * O;
    constexpr int WS_ENTRIES = N * N + 3 * N;
    const int S = tv.S;
    int status = 0;

    if (N > 0 && S > 1) {
        R W[NS][NS], z[NS][3];
#pragma unroll
        for (int r = 0; r < N; ++r) {
#pragma unroll
            for (int c = 0; c < N; ++c) W[r][c] = R(0);
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) z[r][ax] = x0[r][ax];
        }
        R Pp[3], Pc[3], Pn[3], Pnn[3] = {R(0), R(0), R(0)};
        load3<IO, R>(tv.wp, Pp);
        load3<IO, R>(tv.wp + 3, Pc);
        load3<IO, R>(tv.wp + 6, Pn);
        R Tn = R(tv.tm[1]), Tnn = R(1);
        int sn = PATH ? tstar[B] : 0, snn = 0;
        SegBlocks<O, R> left, right;
        seg_blocks<O, R, PATH>(R(tv.tm[0]), vw, pw, PATH ? tstar[0] : 0, Pp, Pc, left);
        for (int k = 1; k < S; ++k) {
            if (k + 1 < S) {  // prefetch waypoint k+2, time k+1 (and its sample index)
                load3<IO, R>(tv.wp + 3 * (k + 2), Pnn);
                Tnn = R(tv.tm[k + 1]);
                if (PATH) snn = tstar[(int64_t)(k + 1) * B];
            }
            seg_blocks<O, R, PATH>(Tn, vw, pw, sn, Pc, Pn, right);
            R A[NS][NS], Bm[NS][NS + 3];
#pragma unroll
            for (int r = 0; r < N; ++r) {
#pragma unroll
                for (int c = 0; c < N; ++c) {
                    R v = left.ee[r][c] + right.ss[r][c];
#pragma unroll
                    for (int j = 0; j < N; ++j) v = fma_<R>(-left.se[j][r], W[j][c], v);
                    A[r][c] = v;
                    Bm[r][c] = right.se[r][c];
                }
#pragma unroll
                for (int ax = 0; ax < 3; ++ax) {
                    R v = left.ep0[r] * Pp[ax];
                    v = fma_<R>(left.ep1[r], Pc[ax], v);
                    v = fma_<R>(right.sp0[r], Pc[ax], v);
                    v = fma_<R>(right.sp1[r], Pn[ax], v);
                    if (PATH) v += left.fe[r][ax] + right.fs[r][ax];
#pragma unroll
                    for (int j = 0; j < N; ++j) v = fma_<R>(left.se[j][r], z[j][ax], v);
                    Bm[r][N + ax] = -v;
                }
            }
            const R piv = spd_solve<NS, NS + 3, R>(A, Bm);
            if (!(piv > R(0))) status |= 2;
            R *wk = ws + (int64_t)(k - 1) * WS_ENTRIES * B;
#pragma unroll
            for (int r = 0; r < N; ++r) {
#pragma unroll
                for (int c = 0; c < N; ++c) { W[r][c] = Bm[r][c]; wk[(int64_t)(r * N + c) * B] = W[r][c]; }
#pragma unroll
                for (int ax = 0; ax < 3; ++ax) { z[r][ax] = Bm[r][N + ax]; wk[(int64_t)(N * N + r * 3 + ax) * B] = z[r][ax]; }
            }
            left = right;
            Tn = Tnn;
            sn = snn;
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) { Pp[ax] = Pc[ax]; Pc[ax] = Pn[ax]; Pn[ax] = Pnn[ax]; }
        }
    }

    // back-substitution fused with coefficient recovery (and the deviation metric)
    R xn[NS][3];
#pragma unroll
    for (int r = 0; r < NS; ++r)
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) xn[r][ax] = (r < N) ? xS[r][ax] : R(0);
    R nanacc = R(0);
    R maxdev = R(0);
    // state of segment k = S-1 (requested before the loop), then one segment ahead inside it
    R Tk = R(tv.tm[S - 1]), P0[3], P1[3], wz[WS_ENTRIES > 0 ? WS_ENTRIES : 1];
    load3<IO, R>(tv.wp + 3 * (S - 1), P0);
    load3<IO, R>(tv.wp + 3 * S, P1);
    int sk = (PATH && dev_out) ? tstar[(int64_t)(S - 1) * B] : 0;
    if (N > 0 && S > 1) {
        const R *wk = ws + (int64_t)(S - 2) * WS_ENTRIES * B;
#pragma unroll
        for (int e = 0; e < WS_ENTRIES; ++e) wz[e] = wk[(int64_t)e * B];
    }
    for (int k = S - 1; k >= 0; --k) {
        R Tp = R(1), Pm[3] = {R(0), R(0), R(0)}, wzp[WS_ENTRIES > 0 ? WS_ENTRIES : 1];
        int sp = 0;
        if (k >= 1) {  // prefetch segment k-1: its time, start waypoint, sample index and factors
            Tp = R(tv.tm[k - 1]);
            load3<IO, R>(tv.wp + 3 * (k - 1), Pm);
            if (PATH && dev_out) sp = tstar[(int64_t)(k - 1) * B];
            if (N > 0 && k >= 2) {
                const R *wk = ws + (int64_t)(k - 2) * WS_ENTRIES * B;
#pragma unroll
                for (int e = 0; e < WS_ENTRIES; ++e) wzp[e] = wk[(int64_t)e * B];
            }
        }
        R xk[NS][3];
        if (k == 0 || N == 0) {
#pragma unroll
            for (int r = 0; r < NS; ++r)
#pragma unroll
                for (int ax = 0; ax < 3; ++ax) xk[r][ax] = (r < N) ? x0[r][ax] : R(0);
        } else {
#pragma unroll
            for (int r = 0; r < N; ++r)
#pragma unroll
                for (int ax = 0; ax < 3; ++ax) {
                    R v = wz[N * N + r * 3 + ax];
#pragma unroll
                    for (int c = 0; c < N; ++c) v = fma_<R>(-wz[r * N + c], xn[c][ax], v);
                    xk[r][ax] = v;
                }
        }
        const R T = Tk;
        R tp[O], ip[M];
        tp[0] = R(1);
#pragma unroll
        for (int e = 1; e < O; ++e) tp[e] = tp[e - 1] * T;
        ip[0] = R(1);
        ip[1] = fast_rcp(T);
#pragma unroll
        for (int e = 2; e < M; ++e) ip[e] = ip[e - 1] * ip[1];
        R d2 = R(0), len2 = R(0);
        const R tau = (PATH && dev_out) ? R(sk) * R(0.0625) : R(0);
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
            R d[M], c[M];
            d[0] = P0[ax];
            d[O] = P1[ax];
#pragma unroll
            for (int r = 0; r < N; ++r) { d[r + 1] = xk[r][ax]; d[O + r + 1] = xn[r][ax]; }
            recover_axis<O, R>(d, tp, ip, c);
            IO *dst = tv.co + (int64_t)k * tv.seg_stride + ax * M;
            IO q[M];
#pragma unroll
            for (int i = 0; i < M; ++i) { q[i] = IO(c[i]); nanacc = fma_<R>(R(q[i]), R(0), nanacc); }
            store_row<M, IO>(dst, q);
            if (dev_out) {
                // deviation at the recorded t* (minimum_snap.cpp:596-617); t* = 0 without path penalty
                const R ts = T * tau;
                R v = c[0];
#pragma unroll
                for (int i = 1; i < M; ++i) v = fma_<R>(v, ts, c[i]);
                const R dp = P1[ax] - P0[ax];
                const R L = fma_<R>(tau, dp, P0[ax]);
                d2 = fma_<R>(v - L, v - L, d2);
                len2 = fma_<R>(dp, dp, len2);
            }
        }
        if (dev_out) {
            const R seg_len = sqrt(len2);
            const R ratio = (seg_len > R(1e-6)) ? sqrt(d2) / seg_len : R(0);
            maxdev = ratio > maxdev ? ratio : maxdev;
        }
#pragma unroll
        for (int r = 0; r < NS; ++r)
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) xn[r][ax] = xk[r][ax];
        Tk = Tp;
        sk = sp;
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) { P1[ax] = P0[ax]; P0[ax] = Pm[ax]; }
#pragma unroll
        for (int e = 0; e < WS_ENTRIES; ++e) wz[e] = wzp[e];
    }
    if (!(nanacc == R(0))) status |= 1;
    if (dev_out) *dev_out = (double)maxdev;
    return status;
}

// Picks t*_k = argmax over 17 samples of the squared distance between the pre-solve polynomial
// and the chord (minimum_snap.cpp:408-439; strict '>' so the first maximum wins).
template <int O, typename IO, typename R>
__device__ void pick_tstar(const TrajView<O, IO> &tv, int *tstar, int64_t B) {
    constexpr int M = 2 * O;
    for (int k = 0; k < tv.S; ++k) {
        const R T = R(tv.tm[k]);
        R P0[3], P1[3];
        load3<IO, R>(tv.wp + 3 * k, P0);
        load3<IO, R>(tv.wp + 3 * (k + 1), P1);
        R c[3][M];
#pragma unroll
        for (int ax = 0; ax < 3; ++ax)
#pragma unroll
            for (int i = 0; i < M; ++i) c[ax][i] = R(tv.co[(int64_t)k * tv.seg_stride + ax * M + i]);
        int best = 0;
        R bestd = R(-1);
        for (int s = 0; s <= 16; ++s) {
            const R tt = T * R(s) / R(16);
            R d2 = R(0);
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) {
                R v = c[ax][0];
#pragma unroll
                for (int i = 1; i < M; ++i) v = fma_<R>(v, tt, c[ax][i]);
                const R L = P0[ax] + (tt / T) * (P1[ax] - P0[ax]);
                d2 += (v - L) * (v - L);
            }
            if (d2 > bestd) { bestd = d2; best = s; }
        }
        tstar[(int64_t)k * B] = best;
    }
}

template <int O, typename IO, typename R, bool PATH>
__global__ void __launch_bounds__(64) minsnap_generic_kernel(GenericArgs a) {
    constexpr int N = O - 1;
    constexpr int NS = (O > 1) ? O - 1 : 1;
    constexpr int M = 2 * O;
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.B) return;
    if (a.skip && a.skip[b]) return;  // re-solve loop: this trajectory already converged
    int64_t seg0;
    int S;
    if (a.seg_off) { seg0 = a.seg_off[b]; S = (int)(a.seg_off[b + 1] - seg0); }
    else { seg0 = b * (int64_t)a.S; S = a.S; }
    if (S < 1) { if (a.status) a.status[b] = 0; if (a.max_dev) a.max_dev[b] = 0.0; return; }
    TrajView<O, IO> tv;
    tv.wp = (const IO *)a.wp + (seg0 + b) * 3;
    tv.tm = (const IO *)a.times + seg0;
    if (a.seg_major && !a.seg_off) { tv.co = (IO *)a.coeffs + b * 3 * M; tv.seg_stride = a.B * 3 * M; }
    else { tv.co = (IO *)a.coeffs + seg0 * 3 * M; tv.seg_stride = 3 * M; }
    tv.S = S;
    const IO *bc = (const IO *)a.bc + (a.bc_per_traj ? b * 12 : 0);
    // fixed boundary derivatives (minimum_snap.cpp:527-555): velocity if order>=2,
    // acceleration if order>=3, every higher one is pinned to zero (:225)
    R x0[NS][3], xS[NS][3];
#pragma unroll
    for (int r = 0; r < NS; ++r)
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
            x0[r][ax] = (r < N && r == 0) ? R(bc[0 * 3 + ax]) : (r < N && r == 1) ? R(bc[2 * 3 + ax]) : R(0);
            xS[r][ax] = (r < N && r == 0) ? R(bc[1 * 3 + ax]) : (r < N && r == 1) ? R(bc[3 * 3 + ax]) : R(0);
        }
    R *ws = (R *)a.ws + b;
    int *tstar = a.tstar ? a.tstar + b : nullptr;
    const R pw = (R)a.path_weight;
    const R vw = (R)(a.vw_per ? a.vw_per[b] : a.vel_zero_weight);
    int status = 0;
    double dev = 0.0;
    if (PATH) {
        if (a.tau_mode != 2) {   // re-solve loop, passes 1..10: the pre-solve ignores vel_zero_weight, t* is unchanged
            status |= solve_pass<O, IO, R, false>(tv, x0, xS, R(0), R(0), ws, tstar, a.B, b, nullptr);
            pick_tstar<O, IO, R>(tv, tstar, a.B);
        }
        status = solve_pass<O, IO, R, true>(tv, x0, xS, pw, vw, ws, tstar, a.B, b, &dev);
    } else {
        status = solve_pass<O, IO, R, false>(tv, x0, xS, R(0), vw, ws, tstar, a.B, b, a.max_dev ? &dev : nullptr);
    }
    if (a.status) a.status[b] = status;
    if (a.max_dev) a.max_dev[b] = dev;
}

template <int O, typename IO, typename R> static hipError_t launch_o(const GenericArgs &a, hipStream_t st) {
    if (a.B == 0) return hipSuccess;
    const int threads = 64;  // one wave per workgroup: small ragged buckets still cover every CU
    const int64_t blocks = (a.B + threads - 1) / threads;
    if (a.path_weight > 0.0) hipLaunchKernelGGL((minsnap_generic_kernel<O, IO, R, true>), dim3((unsigned)blocks), dim3(threads), 0, st, a);
    else hipLaunchKernelGGL((minsnap_generic_kernel<O, IO, R, false>), dim3((unsigned)blocks), dim3(threads), 0, st, a);
    return hipGetLastError();
}

template <typename IO, typename R> static hipError_t launch_r(const GenericArgs &a, hipStream_t st) {
    switch (a.order) {
        case 1: return launch_o<1, IO, R>(a, st);
        case 2: return launch_o<2, IO, R>(a, st);
        case 3: return launch_o<3, IO, R>(a, st);
        case 4: return launch_o<4, IO, R>(a, st);
        case 5: return launch_o<5, IO, R>(a, st);
    }
    return hipErrorInvalidValue;
}

// f32 storage computes in f64 unless f32_arith is set (CSP_FLAG_F32_ARITH): pure-f32 arithmetic
// loses 3..5 digits at order 4..5 (tests/test_gpu_parity.py::test_f32_storage).
hipError_t launch_generic(const GenericArgs &a, bool f32, bool f32_arith, hipStream_t st) {
    if (!f32) return launch_r<double, double>(a, st);
    return f32_arith ? launch_r<float, float>(a, st) : launch_r<float, double>(a, st);
}

size_t generic_ws_entries(int order) {
    const int n = order - 1;
    return (size_t)(n * n + 3 * n);
}

}  // namespace csp
