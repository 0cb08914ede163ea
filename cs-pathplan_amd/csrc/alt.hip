// alt.hip -- banded LDL^T for the altitude optimiser's pentadiagonal SPD systems
// (reference: uavPathPlanning.cpp:1575-1827, the only Eigen::SimplicialLDLT call sites).
// One lane per problem: the factorisation of a pentadiagonal matrix is a two-term recurrence, so a
// problem is sequential in its sample index and parallelism comes from the batch.  The matrix is
// never stored: row i's diagonal and two sub-diagonal entries are rebuilt from the neighbouring
// samples while sweeping; per row the sweep keeps (l1, l2, w) in the workspace for the backward pass.
// A single long problem would want a wave-cooperative cyclic reduction (row N4, last in SURVEY.md 8f).
#include "../../include/csp_alt.h"
#include "../../include/csp_minsnap.h"

#include <hip/hip_runtime.h>
#include "minsnap_hoststage.h"
#include <cmath>

namespace {

struct AltArgs {
    const double *a, *xyz;       // a = elev (optimize) or input_z (global smooth)
    const int64_t *off;
    double *out;
    int32_t *solves;
    double *ws;                  // [total][4]: l1, l2, w, active flag
    int64_t B;
    csp_alt_params p;
};

// climb-rate weight of edge (i, i+1) (uavPathPlanning.cpp:1641-1656)
__device__ __forceinline__ double edge_w(const double *xyz, int64_t i, double rate) {
    if (!(rate > 0.0)) return 0.0;
    const double dist = hypot(xyz[(i + 1) * 3] - xyz[i * 3], xyz[(i + 1) * 3 + 1] - xyz[i * 3 + 1]);
    if (dist <= 1e-9) return 0.0;
    const double denom = dist * rate;
    if (denom <= 1e-12) return 0.0;
    return 1.0 / (denom * denom);
}

// Row i of lambda*L^T L + climb-rate terms: diagonal, H[i][i-1], H[i][i-2].
__device__ __forceinline__ void band_row(const double *xyz, int64_t n, int64_t i, double s, double rate, double w_prev,
                                         double w_next, double &diag, double &e, double &f) {
    diag = 0.0; e = 0.0; f = 0.0;
    if (n >= 3 && s > 0.0) {
        const bool in_m = (i - 1 >= 1 && i - 1 <= n - 2), in_0 = (i >= 1 && i <= n - 2), in_p = (i + 1 >= 1 && i + 1 <= n - 2);
        diag += (in_p ? s : 0.0) + (in_0 ? 4.0 * s : 0.0) + (in_m ? s : 0.0);
        e += (in_0 ? -2.0 * s : 0.0) + (in_m ? -2.0 * s : 0.0);   // entries (t, t-1) of t = i and (t+1, t) of t = i-1
        f += in_m ? s : 0.0;                                        // entry (t+1, t-1) of t = i-1
    }
    diag += w_prev + w_next;
    e += -w_prev;
}

// Full-precision reciprocal (hardware seed + two Newton steps): the pivots are sums of positive weights
// plus 1e-8, far from the ranges IEEE division's fix-ups exist for.
__device__ __forceinline__ double rcp64(double x) {
    double r = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-x, r, 1.0);
    return __builtin_fma(r, e, r);
}

// One banded LDL^T solve.  extra_d(i) / rhs(i) supply the problem-specific diagonal and right side.
// Rows are taken in blocks of ROWS: everything a row needs from memory (edge weights, diagonal
// extras, right-hand sides; the stored factors on the way back) does not depend on the recurrence, so
// a block's loads are issued together and the two-term recurrence then runs from registers with one
// reciprocal per row -- instead of a memory round trip and two divisions per row.
template <class D, class Rh>
__device__ __forceinline__ void banded_solve(const double *xyz, double *ws, int64_t n, double s, double rate, D extra_d, Rh rhs) {
    constexpr int ROWS = 8;
    double d1 = 1.0, d2 = 1.0, r1 = 1.0, r2 = 1.0, l1p = 0.0, y1 = 0.0, y2 = 0.0;   // d_{i-1}, d_{i-2}, their reciprocals, l1_{i-1}, y_{i-1}, y_{i-2}
    double w_prev = 0.0;
    for (int64_t i0 = 0; i0 < n; i0 += ROWS) {
        double dg[ROWS], ee[ROWS], ff[ROWS], rr[ROWS], wn[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int64_t i = i0 + r;
            wn[r] = (i + 1 < n) ? edge_w(xyz, i, rate) : 0.0;
            dg[r] = (i < n) ? extra_d(i) + 1e-8 : 1.0;             // tiny regularisation (:1659-1661)
            rr[r] = (i < n) ? rhs(i) : 0.0;
        }
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int64_t i = i0 + r;
            if (i < n) {
                double diag, e, f;
                band_row(xyz, n, i, s, rate, r == 0 ? w_prev : wn[r - 1], wn[r], diag, e, f);
                dg[r] += diag; ee[r] = e; ff[r] = f;
            }
        }
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int64_t i = i0 + r;
            if (i < n) {
                const double l2 = (i >= 2) ? ff[r] * r2 : 0.0;
                const double l1 = (i >= 1) ? (ee[r] - l2 * l1p * d2) * r1 : 0.0;
                const double d = dg[r] - l1 * l1 * d1 - l2 * l2 * d2;
                const double y = rr[r] - l1 * y1 - l2 * y2;
                const double rd = rcp64(d);
                ws[i * 4 + 0] = l1;
                ws[i * 4 + 1] = l2;
                ws[i * 4 + 2] = y * rd;
                d2 = d1; d1 = d; r2 = r1; r1 = rd; l1p = l1; y2 = y1; y1 = y;
            }
        }
        w_prev = wn[ROWS - 1];
    }
    double z1 = 0.0, z2 = 0.0, l1n = 0.0, l2n = 0.0, l2nn = 0.0;  // z_{i+1}, z_{i+2}, l1_{i+1}, l2_{i+1}, l2_{i+2}
    for (int64_t i0 = n - 1; i0 >= 0; i0 -= ROWS) {
        double a0[ROWS], a1[ROWS], a2[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int64_t i = i0 - r;
            a0[r] = (i >= 0) ? ws[i * 4 + 0] : 0.0;
            a1[r] = (i >= 0) ? ws[i * 4 + 1] : 0.0;
            a2[r] = (i >= 0) ? ws[i * 4 + 2] : 0.0;
        }
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int64_t i = i0 - r;
            if (i >= 0) {
                const double z = a2[r] - l1n * z1 - l2nn * z2;
                ws[i * 4 + 2] = z;
                z2 = z1; z1 = z; l2nn = l2n; l1n = a0[r]; l2n = a1[r];
            }
        }
    }
}

__global__ void __launch_bounds__(64) alt_optimize_kernel(AltArgs a) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.B) return;
    const int64_t o = a.off[b], n = a.off[b + 1] - o;
    if (n <= 0) return;
    const double *xyz = a.xyz + o * 3, *elev = a.a + o;
    double *ws = a.ws + o * 4, *out = a.out + o;
    const csp_alt_params p = a.p;
    banded_solve(xyz, ws, n, p.lambda_smooth, p.max_climb_rate,
                 [&](int64_t i) { return isnan(elev[i]) ? 0.0 : p.lambda_follow; },
                 [&](int64_t i) {
                     if (isnan(elev[i])) return 0.0;
                     const double safe_h = elev[i] + p.safe_distance;   // follow terrain + clearance, never pull down (:1633-1639)
                     return p.lambda_follow * fmax(xyz[i * 3 + 2], safe_h);
                 });
    for (int64_t i = 0; i < n; ++i) {                            // post-check z >= elev + safe_distance (:1689-1710)
        double z = ws[i * 4 + 2];
        if (!isnan(elev[i]) && z < elev[i] + p.safe_distance) z = elev[i] + p.safe_distance;
        out[i] = z;
    }
}

__global__ void __launch_bounds__(64) alt_global_smooth_kernel(AltArgs a) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.B) return;
    const int64_t o = a.off[b], n = a.off[b + 1] - o;
    if (n <= 0) { if (a.solves) a.solves[b] = 0; return; }
    const double *xyz = a.xyz + o * 3, *zin = a.a + o;
    double *ws = a.ws + o * 4, *out = a.out + o;
    const csp_alt_params p = a.p;
    for (int64_t i = 0; i < n; ++i) ws[i * 4 + 3] = 0.0;          // active set empty
    int solves = 0;
    for (int iter = 0; iter < 10; ++iter) {
        banded_solve(xyz, ws, n, p.lambda_smooth, p.max_climb_rate,
                     [&](int64_t i) { return (i == 0 || i == n - 1) ? 1e10 + ((i == 0 && i == n - 1) ? 1e10 : 0.0)
                                                                    : (ws[i * 4 + 3] != 0.0 ? 1e8 : 0.0); },
                     [&](int64_t i) { return (i == 0 || i == n - 1) ? (1e10 + ((i == 0 && i == n - 1) ? 1e10 : 0.0)) * zin[i]
                                                                    : (ws[i * 4 + 3] != 0.0 ? 1e8 * zin[i] : 0.0); });
        ++solves;
        bool violation = false;
        for (int64_t i = 0; i < n; ++i)
            if (ws[i * 4 + 2] < zin[i] - 1e-3 && ws[i * 4 + 3] == 0.0) { ws[i * 4 + 3] = 1.0; violation = true; }
        if (!violation) break;
    }
    for (int64_t i = 0; i < n; ++i) out[i] = fmax(ws[i * 4 + 2], zin[i]);
    if (a.solves) a.solves[b] = solves;
}

// ---- few, long problems: one WAVE per problem ------------------------------------------------------
// What a row needs from memory and from `hypot` (edge weights, band entries, diagonal extras,
// right-hand side) does not depend on the recurrence: the 64 lanes prepare 64 rows at once into LDS.
// The two-term recurrence over those 64 rows is then run by every lane redundantly (broadcast LDS
// reads, ~45 instructions per row, no cross-lane traffic); lane r keeps row r's results and the wave
// writes them coalesced.  Same arithmetic per row as banded_solve: identical results.
template <class D, class Rh>
__device__ __forceinline__ void banded_solve_wave(const double *xyz, double *ws, int64_t n, double s, double rate, D extra_d, Rh rhs,
                                                  double *lds, int lane) {
    double d1 = 1.0, d2 = 1.0, r1 = 1.0, r2 = 1.0, l1p = 0.0, y1 = 0.0, y2 = 0.0;
    double w_carry = 0.0;   // edge weight (i0-1, i0)
    for (int64_t i0 = 0; i0 < n; i0 += 64) {
        const int64_t i = i0 + lane;
        const double wn = (i + 1 < n) ? edge_w(xyz, i, rate) : 0.0;
        double wp = __shfl_up(wn, 1, 64);
        if (lane == 0) wp = w_carry;
        double diag = 0.0, e = 0.0, f = 0.0, dgv = 1.0, rv = 0.0;
        if (i < n) {
            band_row(xyz, n, i, s, rate, wp, wn, diag, e, f);
            dgv = diag + (extra_d(i) + 1e-8);
            rv = rhs(i);
        }
        __syncthreads();
        lds[lane] = dgv; lds[64 + lane] = e; lds[128 + lane] = f; lds[192 + lane] = rv;
        __syncthreads();
        double m1 = 0.0, m2 = 0.0, m3 = 0.0;
        const int rows = (int)((n - i0) < 64 ? (n - i0) : 64);
        for (int r = 0; r < rows; ++r) {
            const int64_t ii = i0 + r;
            const double l2 = (ii >= 2) ? lds[128 + r] * r2 : 0.0;
            const double l1 = (ii >= 1) ? (lds[64 + r] - l2 * l1p * d2) * r1 : 0.0;
            const double d = lds[r] - l1 * l1 * d1 - l2 * l2 * d2;
            const double y = lds[192 + r] - l1 * y1 - l2 * y2;
            const double rd = rcp64(d);
            if (lane == r) { m1 = l1; m2 = l2; m3 = y * rd; }
            d2 = d1; d1 = d; r2 = r1; r1 = rd; l1p = l1; y2 = y1; y1 = y;
        }
        if (i < n) { ws[i * 4 + 0] = m1; ws[i * 4 + 1] = m2; ws[i * 4 + 2] = m3; }
        w_carry = __shfl(wn, 63, 64);
    }
    __syncthreads();
    double z1 = 0.0, z2 = 0.0, l1n = 0.0, l2n = 0.0, l2nn = 0.0;
    for (int64_t i0 = n - 1; i0 >= 0; i0 -= 64) {
        const int64_t i = i0 - lane;     // lane r holds row i0 - r
        __syncthreads();
        lds[lane] = (i >= 0) ? ws[i * 4 + 0] : 0.0;
        lds[64 + lane] = (i >= 0) ? ws[i * 4 + 1] : 0.0;
        lds[128 + lane] = (i >= 0) ? ws[i * 4 + 2] : 0.0;
        __syncthreads();
        double mz = 0.0;
        const int rows = (int)((i0 + 1) < 64 ? (i0 + 1) : 64);
        for (int r = 0; r < rows; ++r) {
            const double z = lds[128 + r] - l1n * z1 - l2nn * z2;
            if (lane == r) mz = z;
            z2 = z1; z1 = z; l2nn = l2n; l1n = lds[r]; l2n = lds[64 + r];
        }
        if (i >= 0) ws[i * 4 + 2] = mz;
    }
    __syncthreads();
}

__global__ void __launch_bounds__(64) alt_optimize_wave_kernel(AltArgs a) {
    __shared__ double lds[256];
    const int lane = threadIdx.x;
    const int64_t b = blockIdx.x;
    const int64_t o = a.off[b], n = a.off[b + 1] - o;
    if (n <= 0) return;
    const double *xyz = a.xyz + o * 3, *elev = a.a + o;
    double *ws = a.ws + o * 4, *out = a.out + o;
    const csp_alt_params p = a.p;
    banded_solve_wave(xyz, ws, n, p.lambda_smooth, p.max_climb_rate,
                      [&](int64_t i) { return isnan(elev[i]) ? 0.0 : p.lambda_follow; },
                      [&](int64_t i) {
                          if (isnan(elev[i])) return 0.0;
                          const double safe_h = elev[i] + p.safe_distance;
                          return p.lambda_follow * fmax(xyz[i * 3 + 2], safe_h);
                      }, lds, lane);
    for (int64_t i = lane; i < n; i += 64) {
        double z = ws[i * 4 + 2];
        if (!isnan(elev[i]) && z < elev[i] + p.safe_distance) z = elev[i] + p.safe_distance;
        out[i] = z;
    }
}

__global__ void __launch_bounds__(64) alt_global_smooth_wave_kernel(AltArgs a) {
    __shared__ double lds[256];
    const int lane = threadIdx.x;
    const int64_t b = blockIdx.x;
    const int64_t o = a.off[b], n = a.off[b + 1] - o;
    if (n <= 0) { if (a.solves && lane == 0) a.solves[b] = 0; return; }
    const double *xyz = a.xyz + o * 3, *zin = a.a + o;
    double *ws = a.ws + o * 4, *out = a.out + o;
    const csp_alt_params p = a.p;
    for (int64_t i = lane; i < n; i += 64) ws[i * 4 + 3] = 0.0;
    __syncthreads();
    int solves = 0;
    for (int iter = 0; iter < 10; ++iter) {
        banded_solve_wave(xyz, ws, n, p.lambda_smooth, p.max_climb_rate,
                          [&](int64_t i) { return (i == 0 || i == n - 1) ? 1e10 + ((i == 0 && i == n - 1) ? 1e10 : 0.0)
                                                                         : (ws[i * 4 + 3] != 0.0 ? 1e8 : 0.0); },
                          [&](int64_t i) { return (i == 0 || i == n - 1) ? (1e10 + ((i == 0 && i == n - 1) ? 1e10 : 0.0)) * zin[i]
                                                                         : (ws[i * 4 + 3] != 0.0 ? 1e8 * zin[i] : 0.0); }, lds, lane);
        ++solves;
        bool violation = false;
        for (int64_t i = lane; i < n; i += 64)
            if (ws[i * 4 + 2] < zin[i] - 1e-3 && ws[i * 4 + 3] == 0.0) { ws[i * 4 + 3] = 1.0; violation = true; }
        __syncthreads();
        if (__builtin_amdgcn_ballot_w64(violation) == 0) break;
    }
    for (int64_t i = lane; i < n; i += 64) out[i] = fmax(ws[i * 4 + 2], zin[i]);
    if (a.solves && lane == 0) a.solves[b] = solves;
}


// ---- ONE (or a few) long problems: block cyclic reduction, one workgroup of 1024 threads per problem ----------------------
// The reference's own call pattern is ONE pentadiagonal problem per plan (uavPathPlanning.cpp:1670-1676, :1796-1799), where a
// sequential recurrence -- however it is dealt over lanes -- costs ~0.2 us per row on the device (0.42 ms at n = 2000,
// 3.7 ms at n = 20000, against tens of microseconds for a banded Cholesky on one CPU core).  Taken two rows at a time the matrix is
// block TRIDIAGONAL with 2x2 blocks (A_k x_{k-1} + B_k x_k + A_{k+1}^T x_{k+1} = r_k), and cyclic reduction eliminates every
// other block row per level: log2(n/2) levels forward, as many back, every level parallel over its rows.  Symmetric
// elimination of an SPD matrix in any order is backward stable, so no pivoting is needed.  A block row keeps 13 doubles
// (A, symmetric B, r / x, and the coupling C it had when it was eliminated); they live in LDS when the problem fits (n <=
// ~2800 samples) and in the workspace otherwise.  Different elimination order from the other two mappings: results agree
// with them to rounding (1e-10 relative measured), not bit for bit.
struct CrStore {
    double *p;        // [13][stride]
    int64_t stride;
    __device__ __forceinline__ double &at(int q, int64_t k) const { return p[(int64_t)q * stride + k]; }
};
enum { CR_A00, CR_A01, CR_A10, CR_A11, CR_B00, CR_B01, CR_B11, CR_R0, CR_R1, CR_C00, CR_C01, CR_C10, CR_C11 };

// y = B^-1 [v0 v1] for the symmetric 2x2 block of row j
__device__ __forceinline__ void cr_binv(const CrStore &s, int64_t j, double (&inv)[3]) {
    const double b00 = s.at(CR_B00, j), b01 = s.at(CR_B01, j), b11 = s.at(CR_B11, j);
    const double rd = rcp64(__builtin_fma(b00, b11, -b01 * b01));
    inv[0] = b11 * rd; inv[1] = -b01 * rd; inv[2] = b00 * rd;
}

template <class D, class Rh>
__device__ __forceinline__ void cr_solve(const CrStore &s, const double *xyz, int64_t n, double sm, double rate, D extra_d, Rh rhs, double *xout) {
    const int tid = threadIdx.x, nt = blockDim.x;
    const int64_t N2 = (n + 1) / 2;
    // ---- assembly: block k = rows 2k, 2k+1 (a missing last row is the identity) ----
    for (int64_t k = tid; k < N2; k += nt) {
        double dg[2], e[2], f[2], rr[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int64_t i = 2 * k + h;
            if (i < n) {
                const double wp = i >= 1 ? edge_w(xyz, i - 1, rate) : 0.0, wn = i + 1 < n ? edge_w(xyz, i, rate) : 0.0;
                double diag;
                band_row(xyz, n, i, sm, rate, wp, wn, diag, e[h], f[h]);
                dg[h] = diag + (extra_d(i) + 1e-8);
                rr[h] = rhs(i);
            } else {
                dg[h] = 1.0; e[h] = 0.0; f[h] = 0.0; rr[h] = 0.0;
            }
        }
        // A_k couples (2k, 2k+1) to (2k-2, 2k-1): [[f_2k, e_2k], [0, f_2k+1]];  B_k = [[d_2k, e_2k+1], [e_2k+1, d_2k+1]]
        s.at(CR_A00, k) = k ? f[0] : 0.0; s.at(CR_A01, k) = k ? e[0] : 0.0; s.at(CR_A10, k) = 0.0; s.at(CR_A11, k) = k ? f[1] : 0.0;
        s.at(CR_B00, k) = dg[0]; s.at(CR_B01, k) = e[1]; s.at(CR_B11, k) = dg[1];
        s.at(CR_R0, k) = rr[0]; s.at(CR_R1, k) = rr[1];
    }
    __syncthreads();
    // ---- forward: at stride st the rows that are odd multiples of st leave; the even multiples absorb them ----
    int64_t st = 1;
    for (; st < N2; st <<= 1) {
        for (int64_t i = (int64_t)tid * 2 * st; i < N2; i += (int64_t)nt * 2 * st) {
            const int64_t jl = i - st, jr = i + st;
            double b00 = s.at(CR_B00, i), b01 = s.at(CR_B01, i), b11 = s.at(CR_B11, i), r0 = s.at(CR_R0, i), r1 = s.at(CR_R1, i);
            double na[4] = {0.0, 0.0, 0.0, 0.0};
            if (jl >= 0) {
                const double a00 = s.at(CR_A00, i), a01 = s.at(CR_A01, i), a10 = s.at(CR_A10, i), a11 = s.at(CR_A11, i);   // A_i: i <- jl
                // the leaving row jl keeps its coupling to me: C_jl = A_i^T
                s.at(CR_C00, jl) = a00; s.at(CR_C01, jl) = a10; s.at(CR_C10, jl) = a01; s.at(CR_C11, jl) = a11;
                double iv[3];
                cr_binv(s, jl, iv);
                // G = A_i B_jl^-1
                const double g00 = a00 * iv[0] + a01 * iv[1], g01 = a00 * iv[1] + a01 * iv[2];
                const double g10 = a10 * iv[0] + a11 * iv[1], g11 = a10 * iv[1] + a11 * iv[2];
                // B_i -= G A_i^T ; r_i -= G r_jl ; A'_i = -G A_jl
                b00 -= g00 * a00 + g01 * a01; b01 -= g00 * a10 + g01 * a11; b11 -= g10 * a10 + g11 * a11;
                const double q0 = s.at(CR_R0, jl), q1 = s.at(CR_R1, jl);
                r0 -= g00 * q0 + g01 * q1; r1 -= g10 * q0 + g11 * q1;
                if (jl - st >= 0) {
                    const double c00 = s.at(CR_A00, jl), c01 = s.at(CR_A01, jl), c10 = s.at(CR_A10, jl), c11 = s.at(CR_A11, jl);
                    na[0] = -(g00 * c00 + g01 * c10); na[1] = -(g00 * c01 + g01 * c11);
                    na[2] = -(g10 * c00 + g11 * c10); na[3] = -(g10 * c01 + g11 * c11);
                }
            }
            if (jr < N2) {
                const double a00 = s.at(CR_A00, jr), a01 = s.at(CR_A01, jr), a10 = s.at(CR_A10, jr), a11 = s.at(CR_A11, jr);   // A_jr: jr <- i
                double iv[3];
                cr_binv(s, jr, iv);
                // G = A_jr^T B_jr^-1 ; B_i -= G A_jr ; r_i -= G r_jr
                const double g00 = a00 * iv[0] + a10 * iv[1], g01 = a00 * iv[1] + a10 * iv[2];
                const double g10 = a01 * iv[0] + a11 * iv[1], g11 = a01 * iv[1] + a11 * iv[2];
                b00 -= g00 * a00 + g01 * a10; b01 -= g00 * a01 + g01 * a11; b11 -= g10 * a01 + g11 * a11;
                const double q0 = s.at(CR_R0, jr), q1 = s.at(CR_R1, jr);
                r0 -= g00 * q0 + g01 * q1; r1 -= g10 * q0 + g11 * q1;
            }
            s.at(CR_B00, i) = b00; s.at(CR_B01, i) = b01; s.at(CR_B11, i) = b11; s.at(CR_R0, i) = r0; s.at(CR_R1, i) = r1;
            s.at(CR_A00, i) = na[0]; s.at(CR_A01, i) = na[1]; s.at(CR_A10, i) = na[2]; s.at(CR_A11, i) = na[3];
        }
        __syncthreads();
    }
    // ---- the last row standing ----
    if (tid == 0) {
        double iv[3];
        cr_binv(s, 0, iv);
        const double q0 = s.at(CR_R0, 0), q1 = s.at(CR_R1, 0);
        s.at(CR_R0, 0) = iv[0] * q0 + iv[1] * q1; s.at(CR_R1, 0) = iv[1] * q0 + iv[2] * q1;
    }
    __syncthreads();
    // ---- back-substitution: the rows that left at stride st, from their two neighbours' solutions (kept in R) ----
    for (st >>= 1; st >= 1; st >>= 1) {
        for (int64_t j = st + (int64_t)tid * 2 * st; j < N2; j += (int64_t)nt * 2 * st) {
            double r0 = s.at(CR_R0, j), r1 = s.at(CR_R1, j);
            {   // x_{j-st}: always exists
                const double x0 = s.at(CR_R0, j - st), x1 = s.at(CR_R1, j - st);
                r0 -= s.at(CR_A00, j) * x0 + s.at(CR_A01, j) * x1; r1 -= s.at(CR_A10, j) * x0 + s.at(CR_A11, j) * x1;
            }
            if (j + st < N2) {
                const double x0 = s.at(CR_R0, j + st), x1 = s.at(CR_R1, j + st);
                r0 -= s.at(CR_C00, j) * x0 + s.at(CR_C01, j) * x1; r1 -= s.at(CR_C10, j) * x0 + s.at(CR_C11, j) * x1;
            }
            double iv[3];
            cr_binv(s, j, iv);
            s.at(CR_R0, j) = iv[0] * r0 + iv[1] * r1; s.at(CR_R1, j) = iv[1] * r0 + iv[2] * r1;
        }
        __syncthreads();
    }
    for (int64_t i = tid; i < n; i += nt) xout[i] = s.at((i & 1) ? CR_R1 : CR_R0, i >> 1);
    __syncthreads();
}

constexpr int kCrThreads = 1024;
constexpr size_t kCrLdsBytes = 148 * 1024;   // dynamic LDS of the cyclic-reduction kernels (a problem of <= ~2900 samples fits)

__device__ __forceinline__ CrStore cr_store(double *lds, double *ws_cr, int64_t n) {
    const int64_t N2 = (n + 1) / 2;
    const bool fits = (size_t)N2 * 13 * 8 <= kCrLdsBytes;
    return CrStore{fits ? lds : ws_cr, N2};
}

// workspace of problem b for the cyclic-reduction kernels: 14 doubles per sample (13 per block row = 6.5 per sample, the
// solution, the active flags)
__global__ void __launch_bounds__(kCrThreads) alt_optimize_cr_kernel(AltArgs a) {
    extern __shared__ __attribute__((aligned(16))) double dyn_lds[];
    const int64_t b = blockIdx.x;
    const int64_t o = a.off[b], n = a.off[b + 1] - o;
    if (n <= 0) return;
    const double *xyz = a.xyz + o * 3, *elev = a.a + o;
    double *ws = a.ws + o * 14, *out = a.out + o;
    double *x = ws, *store = ws + 2 * ((n + 1) / 2) + 2;
    const csp_alt_params p = a.p;
    cr_solve(cr_store(dyn_lds, store, n), xyz, n, p.lambda_smooth, p.max_climb_rate,
             [&](int64_t i) { return isnan(elev[i]) ? 0.0 : p.lambda_follow; },
             [&](int64_t i) {
                 if (isnan(elev[i])) return 0.0;
                 const double safe_h = elev[i] + p.safe_distance;
                 return p.lambda_follow * fmax(xyz[i * 3 + 2], safe_h);
             }, x);
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
        double z = x[i];
        if (!isnan(elev[i]) && z < elev[i] + p.safe_distance) z = elev[i] + p.safe_distance;
        out[i] = z;
    }
}

__global__ void __launch_bounds__(kCrThreads) alt_global_smooth_cr_kernel(AltArgs a) {
    extern __shared__ __attribute__((aligned(16))) double dyn_lds[];
    __shared__ int any_violation;
    const int64_t b = blockIdx.x;
    const int64_t o = a.off[b], n = a.off[b + 1] - o;
    if (n <= 0) { if (a.solves && threadIdx.x == 0) a.solves[b] = 0; return; }
    const double *xyz = a.xyz + o * 3, *zin = a.a + o;
    double *ws = a.ws + o * 14, *out = a.out + o;
    const int64_t N2 = (n + 1) / 2;
    double *x = ws, *act = ws + 2 * N2 + 2 + 13 * N2;     // solution | block rows | active flags
    double *store = ws + 2 * N2 + 2;
    const csp_alt_params p = a.p;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) act[i] = 0.0;
    __syncthreads();
    int solves = 0;
    for (int iter = 0; iter < 10; ++iter) {
        cr_solve(cr_store(dyn_lds, store, n), xyz, n, p.lambda_smooth, p.max_climb_rate,
                 [&](int64_t i) { return (i == 0 || i == n - 1) ? 1e10 + ((i == 0 && i == n - 1) ? 1e10 : 0.0) : (act[i] != 0.0 ? 1e8 : 0.0); },
                 [&](int64_t i) { return (i == 0 || i == n - 1) ? (1e10 + ((i == 0 && i == n - 1) ? 1e10 : 0.0)) * zin[i]
                                                                : (act[i] != 0.0 ? 1e8 * zin[i] : 0.0); }, x);
        ++solves;
        if (threadIdx.x == 0) any_violation = 0;
        __syncthreads();
        bool violation = false;
        for (int64_t i = threadIdx.x; i < n; i += blockDim.x)
            if (x[i] < zin[i] - 1e-3 && act[i] == 0.0) { act[i] = 1.0; violation = true; }
        if (violation) any_violation = 1;
        __syncthreads();
        const int v = any_violation;
        __syncthreads();
        if (!v) break;
    }
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) out[i] = fmax(x[i], zin[i]);
    if (a.solves && threadIdx.x == 0) a.solves[b] = solves;
}

// fewer problems than this: one wave per problem (the lanes prepare 64 rows at once); more: one lane per problem
constexpr int64_t kWaveBatch = 2048;

// a few LONG problems (the reference's own call: one): block cyclic reduction, one 1024-thread workgroup per problem
constexpr int64_t kCrMaxBatch = 64, kCrMinAvgSamples = 512;
inline bool use_cr(int64_t batch, int64_t total) { return batch <= kCrMaxBatch && total >= batch * kCrMinAvgSamples; }

template <typename KC>
hipError_t launch_cr(KC cr_kernel, const AltArgs &a, hipStream_t st) {
    static bool attr_done = false;   // > 64 KB of dynamic LDS needs the opt-in, once per kernel (this function is one per kernel)
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(cr_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kCrLdsBytes);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL(cr_kernel, dim3((unsigned)a.B), dim3(kCrThreads), kCrLdsBytes, st, a);
    return hipGetLastError();
}

template <typename K, typename KW, typename KC>
int run(K kernel, KW wave_kernel, KC cr_kernel, const double *a0, const double *xyz, const int64_t *offsets, int64_t batch, const csp_alt_params *p,
        double *out, int32_t *solves, void *workspace, size_t workspace_bytes, uint32_t mem_space, int32_t device_id, void *stream) {
    if (batch < 0 || !p || (batch > 0 && (!a0 || !xyz || !offsets || !out))) return CSP_ERR_INVALID_ARG;
    if (batch == 0) return CSP_OK;
    if (csp_minsnap_device_count() < 1) return CSP_ERR_NO_DEVICE;
    if (device_id >= 0 && hipSetDevice(device_id) != hipSuccess) return CSP_ERR_HIP;
    hipStream_t st = (hipStream_t)stream;
    AltArgs a;
    a.B = batch; a.p = *p;
    const unsigned blocks = (unsigned)((batch + 63) / 64);
    if (mem_space == CSP_MEM_DEVICE) {
        int64_t total = 0;
        if (hipMemcpyAsync(&total, offsets + batch, 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return CSP_ERR_HIP;
        if (!workspace || workspace_bytes < csp_alt_workspace_bytes(total)) return CSP_ERR_WORKSPACE;
        a.a = a0; a.xyz = xyz; a.off = offsets; a.out = out; a.solves = solves; a.ws = (double *)workspace;
        if (use_cr(batch, total)) return launch_cr(cr_kernel, a, st) == hipSuccess ? CSP_OK : CSP_ERR_HIP;
        if (batch < kWaveBatch) hipLaunchKernelGGL(wave_kernel, dim3((unsigned)batch), dim3(64), 0, st, a);
        else hipLaunchKernelGGL(kernel, dim3(blocks), dim3(64), 0, st, a);
        return hipGetLastError() == hipSuccess ? CSP_OK : CSP_ERR_HIP;
    }
    // host memory: through the device's cached staging arena (minsnap_hoststage.h), synchronous
    const int64_t total = offsets[batch];
    int cur = 0;
    (void)hipGetDevice(&cur);
    csp::HostCall hc(cur, st);
    const size_t o_a = hc.in(a0, (size_t)total * 8), o_xyz = hc.in(xyz, (size_t)total * 24), o_off = hc.in(offsets, (size_t)(batch + 1) * 8);
    const size_t o_out = hc.out(out, (size_t)total * 8), o_sv = hc.out(solves, (size_t)batch * 4);
    const size_t o_ws = hc.scratch(csp_alt_workspace_bytes(total) + 8);   // (offsets of a problem's region: o * 14 doubles)
    if (hc.upload() != hipSuccess) return CSP_ERR_HIP;
    a.a = hc.ptr<const double>(o_a); a.xyz = hc.ptr<const double>(o_xyz); a.off = hc.ptr<const int64_t>(o_off);
    a.out = hc.ptr<double>(o_out); a.solves = hc.ptr<int32_t>(o_sv); a.ws = hc.ptr<double>(o_ws);
    if (use_cr(batch, total)) { if (launch_cr(cr_kernel, a, st) != hipSuccess) return CSP_ERR_HIP; }
    else if (batch < kWaveBatch) hipLaunchKernelGGL(wave_kernel, dim3((unsigned)batch), dim3(64), 0, st, a);
    else hipLaunchKernelGGL(kernel, dim3(blocks), dim3(64), 0, st, a);
    if (hipGetLastError() != hipSuccess || hc.download() != hipSuccess) return CSP_ERR_HIP;
    return CSP_OK;
}

}  // namespace

// 4 doubles per sample for the recurrence kernels; the cyclic-reduction kernels (a few long problems) keep 13 doubles per
// PAIR of samples, the solution and the active flags: 14 doubles per sample + slack covers both
extern "C" size_t csp_alt_workspace_bytes(int64_t total_points) { return total_points > 0 ? (size_t)total_points * 14 * 8 + 4096 : 0; }

extern "C" int csp_alt_optimize_heights_batch(const double *xyz, const double *elev, const int64_t *offsets, int64_t batch,
                                              const csp_alt_params *params, double *out_z, void *workspace,
                                              size_t workspace_bytes, uint32_t mem_space, int32_t device_id, void *hip_stream) {
    return run(alt_optimize_kernel, alt_optimize_wave_kernel, alt_optimize_cr_kernel, elev, xyz, offsets, batch, params, out_z, nullptr, workspace, workspace_bytes, mem_space,
               device_id, hip_stream);
}

extern "C" int csp_alt_global_smooth_batch(const double *input_z, const double *xyz, const int64_t *offsets, int64_t batch,
                                           const csp_alt_params *params, double *out_z, int32_t *solves, void *workspace,
                                           size_t workspace_bytes, uint32_t mem_space, int32_t device_id, void *hip_stream) {
    return run(alt_global_smooth_kernel, alt_global_smooth_wave_kernel, alt_global_smooth_cr_kernel, input_z, xyz, offsets, batch, params, out_z, solves, workspace, workspace_bytes,
               mem_space, device_id, hip_stream);
}
