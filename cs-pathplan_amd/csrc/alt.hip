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

// fewer problems than this: one wave per problem (the lanes prepare 64 rows at once); more: one lane per problem
constexpr int64_t kWaveBatch = 2048;

template <typename K, typename KW>
int run(K kernel, KW wave_kernel, const double *a0, const double *xyz, const int64_t *offsets, int64_t batch, const csp_alt_params *p,
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
    const size_t o_ws = hc.scratch(csp_alt_workspace_bytes(total) + 8);
    if (hc.upload() != hipSuccess) return CSP_ERR_HIP;
    a.a = hc.ptr<const double>(o_a); a.xyz = hc.ptr<const double>(o_xyz); a.off = hc.ptr<const int64_t>(o_off);
    a.out = hc.ptr<double>(o_out); a.solves = hc.ptr<int32_t>(o_sv); a.ws = hc.ptr<double>(o_ws);
    if (batch < kWaveBatch) hipLaunchKernelGGL(wave_kernel, dim3((unsigned)batch), dim3(64), 0, st, a);
    else hipLaunchKernelGGL(kernel, dim3(blocks), dim3(64), 0, st, a);
    if (hipGetLastError() != hipSuccess || hc.download() != hipSuccess) return CSP_ERR_HIP;
    return CSP_OK;
}

}  // namespace

extern "C" size_t csp_alt_workspace_bytes(int64_t total_points) { return total_points > 0 ? (size_t)total_points * 32 : 0; }

extern "C" int csp_alt_optimize_heights_batch(const double *xyz, const double *elev, const int64_t *offsets, int64_t batch,
                                              const csp_alt_params *params, double *out_z, void *workspace,
                                              size_t workspace_bytes, uint32_t mem_space, int32_t device_id, void *hip_stream) {
    return run(alt_optimize_kernel, alt_optimize_wave_kernel, elev, xyz, offsets, batch, params, out_z, nullptr, workspace, workspace_bytes, mem_space,
               device_id, hip_stream);
}

extern "C" int csp_alt_global_smooth_batch(const double *input_z, const double *xyz, const int64_t *offsets, int64_t batch,
                                           const csp_alt_params *params, double *out_z, int32_t *solves, void *workspace,
                                           size_t workspace_bytes, uint32_t mem_space, int32_t device_id, void *hip_stream) {
    return run(alt_global_smooth_kernel, alt_global_smooth_wave_kernel, input_z, xyz, offsets, batch, params, out_z, solves, workspace, workspace_bytes,
               mem_space, device_id, hip_stream);
}
