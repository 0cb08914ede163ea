// minsnap_device.h -- device-side building blocks of the structured minimum-snap solve (gfx950).
//
// What is computed is the reference's SolveQPClosedForm (math_util/minimum_snap.cpp:227-649);
// HOW is different by construction.  The reference materialises dense M, Q, C_T (N x N,
// N = 2*order*S) and takes 5..13 dense LU inverses per call.  Here nothing dense exists:
//
//   * per segment the endpoint-derivative-space Hessian  Qt(T) = M(T)^-T Q(T) M(T)^-1  is a
//     constant (2o x 2o) table times powers of 1/T (scaling law);
//   * the selection matrix C_T (minimum_snap.cpp:268-310) is index arithmetic: interior waypoint
//     k owns free derivatives r = 1..o-1, shared by segment k-1's end and segment k's start;
//   * R_PP (minimum_snap.cpp:564-566) is therefore block-tridiagonal with (o-1)x(o-1) blocks and
//     is IDENTICAL for the three axes, so one block-LDL^T sweep serves three right-hand sides;
//   * p = M^-1 C_T d (minimum_snap.cpp:582-591) is the constant table G = M(1)^-1 applied per
//     segment with T-power scalings.
#pragma once
#include <hip/hip_runtime.h>
#include "minsnap_tables.h"

namespace csp {

template <int O> struct Tab;
#define CSP_TAB(o)                                                                          \
    template <> struct Tab<o> {                                                             \
        __device__ static constexpr double G(int i, int a) { return tables::G##o[i][a]; }   \
        __device__ static constexpr double QT(int a, int b) { return tables::QT##o[a][b]; } \
        __device__ static constexpr double HW(int s, int a) { return tables::HW##o[s][a]; } \
        __device__ static constexpr double KQ(int t, int i) { return tables::KQ##o[t][i]; } \
    };
CSP_TAB(1) CSP_TAB(2) CSP_TAB(3) CSP_TAB(4) CSP_TAB(5)
#undef CSP_TAB

// Full-precision reciprocal without the IEEE division's range fix-ups: hardware seed + two
// Newton steps (fp64) / one (fp32).  Inputs here are segment times and SPD pivots, far from
// the subnormal/overflow ranges the fix-ups exist for.
__device__ __forceinline__ double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-x, r, 1.0);
    return __builtin_fma(r, e, r);
}
__device__ __forceinline__ float fast_rcp(float x) {
    float r = __builtin_amdgcn_rcpf(x);
    float e = __builtin_fmaf(-x, r, 1.0f);
    return __builtin_fmaf(r, e, r);
}

template <typename R> __device__ __forceinline__ R fma_(R a, R b, R c);
template <> __device__ __forceinline__ double fma_<double>(double a, double b, double c) { return __builtin_fma(a, b, c); }
template <> __device__ __forceinline__ float fma_<float>(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// ---------------------------------------------------------------------------------------------
// Per-segment blocks of Qt(T) restricted to what the block-tridiagonal system needs.
// Index convention inside a segment (rows of the reference's per-segment M block,
// minimum_snap.cpp:255-263): a in [0,o) = derivative a at t=0, a in [o,2o) = derivative a-o at T.
// Free derivatives are r = 1..o-1  ->  array index r-1.
// ---------------------------------------------------------------------------------------------
template <int O, typename R> struct SegBlocks {
    static constexpr int N = (O > 1) ? O - 1 : 1;  // storage extent (O==1 has no free derivative)
    R ss[N][N];   // Qt[start r][start r']
    R se[N][N];   // Qt[start r][end r']    (the coupling block C_k; its transpose is A_{k+1})
    R ee[N][N];   // Qt[end r][end r']
    R sp0[N], sp1[N];  // Qt[start r][start pos], Qt[start r][end pos]
    R ep0[N], ep1[N];  // Qt[end r][start pos],   Qt[end r][end pos]
    R fs[N][3], fe[N][3];  // linear term f~ = M^-T f_coeff at start r / end r (path penalty only)
};

// Hermite weights h[a] with P(t*) = sum_a h[a] d[a] at t* = T*tau  (h = M^-T phi(t*)).
template <int O, typename R>
__device__ __forceinline__ void hermite_weights(R T, R tau, R (&h)[2 * O]) {
    constexpr int M = 2 * O;
    R taup[M];
    taup[0] = R(1);
#pragma unroll
    for (int e = 1; e < M; ++e) taup[e] = taup[e - 1] * tau;
    R tp[O];
    tp[0] = R(1);
#pragma unroll
    for (int e = 1; e < O; ++e) tp[e] = tp[e - 1] * T;
#pragma unroll
    for (int a = 0; a < M; ++a) {
        R acc = R(0);
#pragma unroll
        for (int i = 0; i < M; ++i) acc = fma_<R>(R(Tab<O>::G(i, a)), taup[M - 1 - i], acc);
        h[a] = acc * tp[a % O];
    }
}

// Builds the blocks of one segment.  `vw` is the zero-velocity weight (minimum_snap.cpp:473-509:
// in derivative space it is exactly +w on the two velocity diagonals of the segment), `pw` the
// path weight with its sample index `tau_idx` in 0..16 (t* = T*tau_idx/16, :408-439) and the
// chord point L = P0 + tau*(P1-P0) (:448-460).
template <int O, typename R, bool PATH>
__device__ __forceinline__ void seg_blocks(R T, R vw, R pw, int tau_idx, const R (&p0)[3],
                                           const R (&p1)[3], SegBlocks<O, R> &sb) {
    constexpr int M = 2 * O;
    constexpr int N = O - 1;
    R ip[M];  // ip[e] = T^-e
    ip[0] = R(1);
    ip[1] = fast_rcp(T);
#pragma unroll
    for (int e = 2; e < M; ++e) ip[e] = ip[e - 1] * ip[1];
#define CSP_QT(a, b) (R(Tab<O>::QT(a, b)) * ip[M - 1 - ((a) % O) - ((b) % O)])
#pragma unroll
    for (int r = 0; r < N; ++r) {
#pragma unroll
        for (int c = 0; c < N; ++c) {
            sb.ss[r][c] = CSP_QT(r + 1, c + 1);
            sb.se[r][c] = CSP_QT(r + 1, O + c + 1);
            sb.ee[r][c] = CSP_QT(O + r + 1, O + c + 1);
        }
        sb.sp0[r] = CSP_QT(r + 1, 0);
        sb.sp1[r] = CSP_QT(r + 1, O);
        sb.ep0[r] = CSP_QT(O + r + 1, 0);
        sb.ep1[r] = CSP_QT(O + r + 1, O);
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) { sb.fs[r][ax] = R(0); sb.fe[r][ax] = R(0); }
    }
#undef CSP_QT
    if (N >= 1) { sb.ss[0][0] += vw; sb.ee[0][0] += vw; }
    if (PATH) {
        R h[M];
        const R tau = R(tau_idx) * R(0.0625);
        hermite_weights<O, R>(T, tau, h);
#pragma unroll
        for (int r = 0; r < N; ++r) {
            const R hs = pw * h[r + 1], he = pw * h[O + r + 1];
#pragma unroll
            for (int c = 0; c < N; ++c) {
                sb.ss[r][c] = fma_<R>(hs, h[c + 1], sb.ss[r][c]);
                sb.se[r][c] = fma_<R>(hs, h[O + c + 1], sb.se[r][c]);
                sb.ee[r][c] = fma_<R>(he, h[O + c + 1], sb.ee[r][c]);
            }
            sb.sp0[r] = fma_<R>(hs, h[0], sb.sp0[r]);
            sb.sp1[r] = fma_<R>(hs, h[O], sb.sp1[r]);
            sb.ep0[r] = fma_<R>(he, h[0], sb.ep0[r]);
            sb.ep1[r] = fma_<R>(he, h[O], sb.ep1[r]);
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) {
                const R L = fma_<R>(tau, p1[ax] - p0[ax], p0[ax]);
                // f_coeff = -2*w*phi*L (minimum_snap.cpp:457-459), used UN-halved at :579
                sb.fs[r][ax] = R(-2) * hs * L;
                sb.fe[r][ax] = R(-2) * he * L;
            }
        }
    }
}

// In-place LDL^T solve of a small SPD system A X = B (A symmetric, lower triangle read),
// NR right-hand sides stored as columns of B[N][NR].  Returns the smallest pivot.
template <int N, int NR, typename R>
__device__ __forceinline__ R spd_solve(R (&A)[N][N], R (&B)[N][NR]) {
    R L[N][N];
    R dinv[N];
    R minpiv = R(0);
#pragma unroll
    for (int j = 0; j < N; ++j) {
        R d = A[j][j];
#pragma unroll
        for (int k = 0; k < j; ++k) d = fma_<R>(-L[j][k], A[j][k], d);  // A[j][k] holds L[j][k]*d_k
        minpiv = (j == 0) ? d : (d < minpiv ? d : minpiv);
        dinv[j] = fast_rcp(d);
#pragma unroll
        for (int i = j + 1; i < N; ++i) {
            R v = A[i][j];
#pragma unroll
            for (int k = 0; k < j; ++k) v = fma_<R>(-L[i][k], A[j][k], v);
            A[i][j] = v;             // = L[i][j] * d_j
            L[i][j] = v * dinv[j];
        }
    }
#pragma unroll
    for (int c = 0; c < NR; ++c) {
#pragma unroll
        for (int i = 1; i < N; ++i)
#pragma unroll
            for (int k = 0; k < i; ++k) B[i][c] = fma_<R>(-L[i][k], B[k][c], B[i][c]);
#pragma unroll
        for (int i = 0; i < N; ++i) B[i][c] *= dinv[i];
#pragma unroll
        for (int i = N - 2; i >= 0; --i)
#pragma unroll
            for (int k = i + 1; k < N; ++k) B[i][c] = fma_<R>(-L[k][i], B[k][c], B[i][c]);
    }
    return minpiv;
}

// Monomial coefficients of one segment and one axis from its endpoint derivatives
// d = [derivs 0..o-1 at t=0 ; derivs 0..o-1 at t=T]:  c = diag(T^-pow) G diag(T^deriv) d
// (= M(T)^-1 d, minimum_snap.cpp:585-591).  `ip[e]` = T^-e, `tp[e]` = T^e.
template <int O, typename R>
__device__ __forceinline__ void recover_axis(const R (&d)[2 * O], const R (&tp)[O], const R (&ip)[2 * O],
                                             R (&c)[2 * O]) {
    constexpr int M = 2 * O;
    R dh[M];
#pragma unroll
    for (int a = 0; a < M; ++a) dh[a] = d[a] * tp[a % O];
#pragma unroll
    for (int i = 0; i < M; ++i) {
        R acc = R(0);
#pragma unroll
        for (int a = 0; a < M; ++a) {
            constexpr double zero = 0.0;
            if (Tab<O>::G(i, a) != zero) acc = fma_<R>(R(Tab<O>::G(i, a)), dh[a], acc);
        }
        c[i] = acc * ip[M - 1 - i];
    }
}

}  // namespace csp
