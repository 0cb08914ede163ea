// minsnap_plan.hip -- the parts of TrajectoryGeneratorTool::GenerateTrajectoryMatrix
// (math_util/minimum_snap.cpp:22-206) around the solve, batched:
//   * the re-solve loop's per-trajectory bookkeeping (:76-90): while max_dev > 0.2 and fewer than
//     10 increases, vel_zero_weight <- (w < 1e-6 ? 0.01 : 2w);
//   * polynomial sampling at dt = min(0.1, T/10) with sequential distance thinning, the end-point
//     rule and the climb-rate / turn-radius statistics (:97-195).
// One lane per trajectory: both are inherently sequential per trajectory.
#include "minsnap_launch.h"

namespace csp {

__global__ void __launch_bounds__(256) resolve_init_kernel(double *vw, int32_t *iters, int32_t *done, double vw0, int64_t B) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    vw[b] = vw0;
    iters[b] = 0;
    done[b] = 0;
}

__global__ void __launch_bounds__(256) resolve_update_kernel(const double *max_dev, double *vw, int32_t *iters,
                                                             int32_t *done, int64_t B) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B || done[b]) return;
    if (max_dev[b] > 0.2 && iters[b] < 10) {
        const double w = vw[b];
        vw[b] = (w < 1e-6) ? 0.01 : w * 2.0;
        iters[b] += 1;
    } else {
        done[b] = 1;
    }
}

__global__ void __launch_bounds__(256) fill_f64_kernel(double *p, double v, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
hipError_t launch_fill_f64(double *p, double v, int64_t n, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(fill_f64_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p, v, n);
    return hipGetLastError();
}

hipError_t launch_resolve_init(double *vw, int32_t *iters, int32_t *done, double vw0, int64_t B, hipStream_t st) {
    if (B == 0) return hipSuccess;
    hipLaunchKernelGGL(resolve_init_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, st, vw, iters, done, vw0, B);
    return hipGetLastError();
}
hipError_t launch_resolve_update(const double *max_dev, double *vw, int32_t *iters, int32_t *done, int64_t B, hipStream_t st) {
    if (B == 0) return hipSuccess;
    hipLaunchKernelGGL(resolve_update_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, st, max_dev, vw, iters, done, B);
    return hipGetLastError();
}

template <int M>
__device__ __forceinline__ void eval_poly(const double (&c)[3][M], double t, double (&out)[3]) {
    // sum of c_k * t^(M-1-k), k ascending, like the reference's eval lambda (:104-117)
    double pw[M];
    pw[M - 1] = 1.0;
#pragma unroll
    for (int k = M - 2; k >= 0; --k) pw[k] = pw[k + 1] * t;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        double v = 0.0;
#pragma unroll
        for (int k = 0; k < M; ++k) v += c[a][k] * pw[k];
        out[a] = v;
    }
}

template <int O, typename IO>
__global__ void __launch_bounds__(64) sample_kernel(SampleArgs a) {
    constexpr int M = 2 * O;
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.B) return;
    int64_t seg0;
    int S;
    if (a.seg_off) { seg0 = a.seg_off[b]; S = (int)(a.seg_off[b + 1] - seg0); }
    else { seg0 = b * (int64_t)a.S; S = a.S; }
    const IO *tm = (const IO *)a.times + seg0;
    const IO *co;
    int64_t seg_stride;
    if (a.seg_major && !a.seg_off) { co = (const IO *)a.coeffs + b * 3 * M; seg_stride = a.B * 3 * M; }
    else { co = (const IO *)a.coeffs + seg0 * 3 * M; seg_stride = 3 * M; }
    IO *out = (IO *)a.samples + b * a.capacity * 3;
    int64_t n = 0;
    double p0[3] = {0, 0, 0}, p1[3] = {0, 0, 0};  // the two most recently recorded samples
    double max_climb = 0.0, min_r = 1.0e12;
    auto record = [&](const double (&p)[3]) {
        if (n < a.capacity) { out[n * 3] = (IO)p[0]; out[n * 3 + 1] = (IO)p[1]; out[n * 3 + 2] = (IO)p[2]; }
        if (n >= 1) {  // statistics over consecutive recorded samples (:167-193)
            const double dx = p[0] - p1[0], dy = p[1] - p1[1], dz = fabs(p[2] - p1[2]);
            const double hd = sqrt(dx * dx + dy * dy);
            if (hd > 1e-6) { const double r = dz / hd; if (r > max_climb) max_climb = r; }
            if (n >= 2) {
                const double u[3] = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]};
                const double w[3] = {p[0] - p0[0], p[1] - p0[1], p[2] - p0[2]};
                const double la = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
                const double lb = sqrt(dx * dx + dy * dy + (p[2] - p1[2]) * (p[2] - p1[2]));
                const double lc = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
                const double cx = u[1] * w[2] - u[2] * w[1], cy = u[2] * w[0] - u[0] * w[2], cz = u[0] * w[1] - u[1] * w[0];
                const double area = 0.5 * sqrt(cx * cx + cy * cy + cz * cz);
                if (area > 1e-8) { const double R = la * lb * lc / (4.0 * area); if (R < min_r) min_r = R; }
            }
        }
        p0[0] = p1[0]; p0[1] = p1[1]; p0[2] = p1[2];
        p1[0] = p[0]; p1[1] = p[1]; p1[2] = p[2];
        ++n;
    };
    double prev[3], cur[3];
    for (int seg = 0; seg < S; ++seg) {
        const IO *rec = co + (int64_t)seg * seg_stride;
        double c[3][M];   // the segment's record, read ONCE (the evaluations below run from registers)
#pragma unroll
        for (int ax = 0; ax < 3; ++ax)
#pragma unroll
            for (int k = 0; k < M; ++k) c[ax][k] = (double)rec[ax * M + k];
        const double T = (double)tm[seg];
        double dt = 0.1;
        if (dt > T / 10.0) dt = T / 10.0;  // at least 10 evaluations per segment (:126)
        eval_poly<M>(c, 0.0, prev);
        if (n == 0) record(prev);
        for (double t = dt; t <= T + 1e-12; t += dt) {  // accumulated like the reference (:140)
            eval_poly<M>(c, t < T ? t : T, cur);
            const double dx = cur[0] - prev[0], dy = cur[1] - prev[1], dz = cur[2] - prev[2];
            if (sqrt(dx * dx + dy * dy + dz * dz) >= a.sample_distance) {
                prev[0] = cur[0]; prev[1] = cur[1]; prev[2] = cur[2];
                record(cur);
            }
        }
        if (seg == S - 1) {  // make sure the end point is present, without duplicating it (:157-160)
            eval_poly<M>(c, T, cur);
            const double dx = p1[0] - cur[0], dy = p1[1] - cur[1], dz = p1[2] - cur[2];
            if (n == 0 || sqrt(dx * dx + dy * dy + dz * dz) > 1e-6) record(cur);
        }
    }
    a.counts[b] = (int32_t)n;
    if (a.stats) { a.stats[b * 2] = max_climb; a.stats[b * 2 + 1] = min_r; }
}

template <typename IO> static hipError_t launch_sample_t(const SampleArgs &a, hipStream_t st) {
    const dim3 grid((unsigned)((a.B + 63) / 64)), block(64);
    switch (a.order) {
        case 1: hipLaunchKernelGGL((sample_kernel<1, IO>), grid, block, 0, st, a); break;
        case 2: hipLaunchKernelGGL((sample_kernel<2, IO>), grid, block, 0, st, a); break;
        case 3: hipLaunchKernelGGL((sample_kernel<3, IO>), grid, block, 0, st, a); break;
        case 4: hipLaunchKernelGGL((sample_kernel<4, IO>), grid, block, 0, st, a); break;
        case 5: hipLaunchKernelGGL((sample_kernel<5, IO>), grid, block, 0, st, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_sample(const SampleArgs &a, bool f32, hipStream_t st) {
    if (a.B == 0) return hipSuccess;
    return f32 ? launch_sample_t<float>(a, st) : launch_sample_t<double>(a, st);
}

}  // namespace csp
