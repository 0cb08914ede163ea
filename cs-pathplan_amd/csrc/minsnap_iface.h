// minsnap_iface.h -- what the multi-lane kernels (minsnap_chunked.hip, minsnap_span.hip) share: the LDS
// image of the interface system left by the chunk / span Schur complements, its twisted solve, and
// the per-axis record stores.
#pragma once
#include "minsnap_fixed_impl.h"

namespace csp {
namespace iface {

using fixedk::SmallSpd;

template <typename IO, int M> __device__ __forceinline__ void store_axis(IO *dst, const double (&c)[M]);
template <> __device__ __forceinline__ void store_axis<double, 4>(double *d, const double (&c)[4]) {
    reinterpret_cast<double2 *>(d)[0] = make_double2(c[0], c[1]);
    reinterpret_cast<double2 *>(d)[1] = make_double2(c[2], c[3]);
}
template <> __device__ __forceinline__ void store_axis<double, 6>(double *d, const double (&c)[6]) {
#pragma unroll
    for (int i = 0; i < 3; ++i) reinterpret_cast<double2 *>(d)[i] = make_double2(c[2 * i], c[2 * i + 1]);
}
template <> __device__ __forceinline__ void store_axis<double, 8>(double *d, const double (&c)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) reinterpret_cast<double2 *>(d)[i] = make_double2(c[2 * i], c[2 * i + 1]);
}
template <> __device__ __forceinline__ void store_axis<double, 10>(double *d, const double (&c)[10]) {
#pragma unroll
    for (int i = 0; i < 5; ++i) reinterpret_cast<double2 *>(d)[i] = make_double2(c[2 * i], c[2 * i + 1]);
}
template <> __device__ __forceinline__ void store_axis<float, 4>(float *d, const double (&c)[4]) {
    reinterpret_cast<float4 *>(d)[0] = make_float4((float)c[0], (float)c[1], (float)c[2], (float)c[3]);
}
template <> __device__ __forceinline__ void store_axis<float, 6>(float *d, const double (&c)[6]) {
#pragma unroll
    for (int i = 0; i < 3; ++i) reinterpret_cast<float2 *>(d)[i] = make_float2((float)c[2 * i], (float)c[2 * i + 1]);
}
template <> __device__ __forceinline__ void store_axis<float, 8>(float *d, const double (&c)[8]) {
    reinterpret_cast<float4 *>(d)[0] = make_float4((float)c[0], (float)c[1], (float)c[2], (float)c[3]);
    reinterpret_cast<float4 *>(d)[1] = make_float4((float)c[4], (float)c[5], (float)c[6], (float)c[7]);
}
template <> __device__ __forceinline__ void store_axis<float, 10>(float *d, const double (&c)[10]) {
#pragma unroll
    for (int i = 0; i < 5; ++i) reinterpret_cast<float2 *>(d)[i] = make_float2((float)c[2 * i], (float)c[2 * i + 1]);
}

// LDS image of the interface system, one slot per lane: [entry][64] doubles.  Interface i (between chunk
// i-1 and chunk i of a trajectory) lives in the slot of chunk i's lane: D (lower triangle, both
// neighbours' parts summed), the right-hand side r, and chunk i's coupling block E.
template <int O> struct IfaceLds {
    static constexpr int N = O - 1;
    static constexpr int ND = N * (N + 1) / 2;   // D lower triangle
    static constexpr int OFF_R = ND, OFF_E = ND + 3 * N, ENTRIES = ND + 3 * N + N * N;
    // the span kernel later stages one fp64 record (6*O doubles) + 16 bytes of padding per lane here
    static constexpr int STAGE_DOUBLES = 64 * (ENTRIES > 6 * O + 2 ? ENTRIES : 6 * O + 2);
};

// Twisted elimination of the interface system.  Lane j (chunk j of nch) solves interface j+1 -- its
// right end -- from the Schur carry of interfaces 1..j on the left and nch-1..j+2 on the right, and
// takes its left end from lane j-1 through `xch`.  x0 / xn: the trajectory's known end derivatives.
// nch-2 block steps for every lane, so the loop is wave-uniform.  Called by all 64 lanes.
template <int O>
__device__ __forceinline__ void iface_solve(const double *lds, double *xch, int lane, int j, int nch, bool active,
                                            const double (&x0)[O - 1][3], const double (&xn)[O - 1][3],
                                            double (&xL)[O - 1][3], double (&xR)[O - 1][3], bool &spd) {
    constexpr int N = O - 1;
    using IL = IfaceLds<O>;
    const bool solver = active && j + 1 < nch;
    {
        const int base = lane - j;
        const int nst = solver ? nch - 2 : -1;   // steps of this lane (-1: takes no part)
        double cS[N][N], cr[N][3];     // current Schur carry onto the next interface
        double lS[N][N], lr[N][3];     // the finished left carry (onto interface j+1)
        double rinit[N][3];
#pragma unroll
        for (int r = 0; r < N; ++r) {
#pragma unroll
            for (int qq = 0; qq < N; ++qq) { cS[r][qq] = 0.0; lS[r][qq] = 0.0; }
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) { cr[r][ax] = 0.0; lr[r][ax] = 0.0; rinit[r][ax] = 0.0; }
        }
        if (solver) {
            // the known ends enter as right-hand sides: -E_0^T x_0 onto interface 1, -E_{nch-1} x_n onto nch-1
#pragma unroll
            for (int r = 0; r < N; ++r)
#pragma unroll
                for (int qq = 0; qq < N; ++qq) {
                    const double e0 = lds[(IL::OFF_E + qq * N + r) * 64 + base];               // E_0[qq][r]
                    const double en = lds[(IL::OFF_E + r * N + qq) * 64 + base + nch - 1];     // E_{nch-1}[r][qq]
#pragma unroll
                    for (int ax = 0; ax < 3; ++ax) {
                        cr[r][ax] = __builtin_fma(-e0, x0[qq][ax], cr[r][ax]);
                        rinit[r][ax] = __builtin_fma(-en, xn[qq][ax], rinit[r][ax]);
                    }
                }
        }
        for (int t = 0; __builtin_amdgcn_ballot_w64(t <= nst) != 0; ++t) {
            if (t == j) {   // the left sweep has reached my left interface: keep its carry, start from the right end
#pragma unroll
                for (int r = 0; r < N; ++r) {
#pragma unroll
                    for (int qq = 0; qq <= r; ++qq) { lS[r][qq] = cS[r][qq]; cS[r][qq] = 0.0; }
#pragma unroll
                    for (int ax = 0; ax < 3; ++ax) { lr[r][ax] = cr[r][ax]; cr[r][ax] = rinit[r][ax]; }
                }
            }
            if (t < nst) {
                const bool isleft = t < j;
                const int i = isleft ? t + 1 : nch - 1 - (t - j);
                const int slotD = base + i, slotE = base + (isleft ? i : i - 1);
                const int sr = isleft ? N : 1, sc = isleft ? 1 : N;   // F = E_i (left) or E_{i-1}^T (right)
                double Sm[N][N], rc[N][3], Wf[N][N], F[N][N];
                int e = 0;
#pragma unroll
                for (int r = 0; r < N; ++r)
#pragma unroll
                    for (int qq = 0; qq <= r; ++qq) Sm[r][qq] = lds[(e++) * 64 + slotD] + cS[r][qq];
#pragma unroll
                for (int r = 0; r < N; ++r)
#pragma unroll
                    for (int ax = 0; ax < 3; ++ax) rc[r][ax] = lds[(e++) * 64 + slotD] + cr[r][ax];
#pragma unroll
                for (int r = 0; r < N; ++r)
#pragma unroll
                    for (int qq = 0; qq < N; ++qq) { F[r][qq] = lds[(IL::OFF_E + r * sr + qq * sc) * 64 + slotE]; Wf[r][qq] = F[r][qq]; }
                spd &= SmallSpd<N, N>::solve(Sm, Wf);   // Wf = S^-1 F
                // carry onto the next interface: -F^T S^-1 F and -F^T S^-1 rc = -(S^-1 F)^T rc (S symmetric)
#pragma unroll
                for (int r = 0; r < N; ++r) {
#pragma unroll
                    for (int qq = 0; qq <= r; ++qq) {
                        double v = 0.0;
#pragma unroll
                        for (int k = 0; k < N; ++k) v = __builtin_fma(-F[k][r], Wf[k][qq], v);
                        cS[r][qq] = v;
                    }
#pragma unroll
                    for (int ax = 0; ax < 3; ++ax) {
                        double v = 0.0;
#pragma unroll
                        for (int k = 0; k < N; ++k) v = __builtin_fma(-Wf[k][r], rc[k][ax], v);
                        cr[r][ax] = v;
                    }
                }
            }
        }
        // both carries now sit on interface j+1
#pragma unroll
        for (int r = 0; r < N; ++r)
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) xR[r][ax] = xn[r][ax];
        if (solver) {
            const int slotD = base + j + 1;
            double Sm[N][N], R[N][3];
            int e = 0;
#pragma unroll
            for (int r = 0; r < N; ++r)
#pragma unroll
                for (int qq = 0; qq <= r; ++qq) Sm[r][qq] = lds[(e++) * 64 + slotD] + lS[r][qq] + cS[r][qq];
#pragma unroll
            for (int r = 0; r < N; ++r)
#pragma unroll
                for (int ax = 0; ax < 3; ++ax) R[r][ax] = lds[(e++) * 64 + slotD] + lr[r][ax] + cr[r][ax];
            spd &= SmallSpd<N, 3>::solve(Sm, R);
#pragma unroll
            for (int r = 0; r < N; ++r)
#pragma unroll
                for (int ax = 0; ax < 3; ++ax) {
                    xR[r][ax] = R[r][ax];
                    xch[(r * 3 + ax) * 64 + lane] = R[r][ax];
                }
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < N; ++r)
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) xL[r][ax] = (active && j >= 1) ? xch[(r * 3 + ax) * 64 + lane - 1] : x0[r][ax];

}

}  // namespace iface
}  // namespace csp
