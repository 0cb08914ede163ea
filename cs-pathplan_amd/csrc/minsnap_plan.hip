// minsnap_plan.hip -- the parts of TrajectoryGeneratorTool::GenerateTrajectoryMatrix
// (math_util/minimum_snap.cpp:22-206) around the solve, batched:
//   * the re-solve loop's per-trajectory bookkeeping (:76-90): while max_dev > 0.2 and fewer than
//     10 increases, vel_zero_weight <- (w < 1e-6 ? 0.01 : 2w);
//   * polynomial sampling at dt = min(0.1, T/10) with sequential distance thinning, the end-point
//     rule and the climb-rate / turn-radius statistics (:97-195).
// The loop bookkeeping is one lane per trajectory; sampling runs one lane per (trajectory, segment)
// for trajectories of up to 64 segments and one lane per trajectory beyond.
#include <mutex>

#include "minsnap_timealloc.h"

namespace csp {

__global__ void __launch_bounds__(256) resolve_init_kernel(double *vw, int32_t *iters, int32_t *done, int32_t *pending, double vw0,
                                                           int64_t B) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b == 0 && pending) *pending = 0;
    if (b >= B) return;
    if (vw) vw[b] = vw0;
    iters[b] = 0;
    done[b] = 0;
}

__global__ void __launch_bounds__(256) resolve_update_kernel(const double *max_dev, double *vw, int32_t *iters,
                                                             int32_t *done, int32_t *pending, int64_t B) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B || done[b]) return;
    if (max_dev[b] > 0.2 && iters[b] < 10) {
        const double w = vw[b];
        vw[b] = (w < 1e-6) ? 0.01 : w * 2.0;
        iters[b] += 1;
        if (pending) atomicAdd(pending, 1);   // host-memory callers stop the loop when nobody is left (capi plan_device)
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

hipError_t launch_resolve_init(double *vw, int32_t *iters, int32_t *done, int32_t *pending, double vw0, int64_t B, hipStream_t st) {
    if (B == 0) return hipSuccess;
    hipLaunchKernelGGL(resolve_init_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, st, vw, iters, done, pending, vw0, B);
    return hipGetLastError();
}
hipError_t launch_resolve_update(const double *max_dev, double *vw, int32_t *iters, int32_t *done, int32_t *pending, int64_t B,
                                 hipStream_t st) {
    if (B == 0) return hipSuccess;
    hipLaunchKernelGGL(resolve_update_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, st, max_dev, vw, iters, done, pending, B);
    return hipGetLastError();
}

// Upper bound of the candidate loop `t <= T + 1e-12` (:140).  The loop advances by dt = min(0.1, T/10),
// so its trip count is 10 + 1e-11/T for short segments: T = 0 (two coincident waypoints with
// min_time_s = 0, or a caller-supplied zero), a denormal T or T in (-1e-12, 0) would never leave it --
// the reference spins on the CPU there, on the device that is a hang.  Degenerate segment times
// (T < 1e-14: more than 1000 candidates all clamped to t = T; negative, NaN) and absurd or non-finite
// ones (> 1e7 s) therefore get NO candidates: NaN compares false with every t.  Such a trajectory is
// already flagged by the solve's status (non-finite coefficients or a non-positive pivot).
__device__ __forceinline__ double t_end(double T) { return (T >= 1.0e-14 && T <= 1.0e7) ? T + 1e-12 : __builtin_nan(""); }

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

// The two statistics the reference PRINTS (max climb rate, min turn radius; minimum_snap.cpp:163-195) are an extremum over
// consecutive recorded samples of dz / hd and of the circumradius la lb lc / (4 area).  Both are monotone in their squares, so
// the kernels track the extremum of the SQUARES -- no square root and one reciprocal per quantity and sample instead of five
// fp64 square roots and two divisions -- and take the two square roots once, when the statistics are written (round 3).
// The reference's thresholds carry over (hd > 1e-6 <=> hd^2 > 1e-12, area > 1e-8 <=> |u x w|^2 > 4e-16); the values agree
// with the reference's to rounding (1e-15 relative), which is what the tests ask of a printed diagnostic (1e-6); every
// sampler uses these functions, so they agree with each other bit for bit.
__device__ __forceinline__ double stat_rcp(double x) {   // hardware seed + two Newton steps; arguments are 1e-16 .. 1e20
    double r = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-x, r, 1.0);
    return __builtin_fma(r, e, r);
}
__device__ __forceinline__ void stat_look(const double (&p0)[3], const double (&p1)[3], const double (&p)[3], bool have1, bool have2,
                                          double &max_climb2, double &min_r2) {
    if (have1) {
        const double dx = p[0] - p1[0], dy = p[1] - p1[1], dz = p[2] - p1[2];
        const double hd2 = dx * dx + dy * dy;
        if (hd2 > 1e-12) { const double r2 = dz * dz * stat_rcp(hd2); if (r2 > max_climb2) max_climb2 = r2; }
        if (have2) {
            const double u[3] = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]};
            const double w[3] = {p[0] - p0[0], p[1] - p0[1], p[2] - p0[2]};
            const double la2 = u[0] * u[0] + u[1] * u[1] + u[2] * u[2];
            const double lb2 = hd2 + dz * dz;
            const double lc2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
            const double cx = u[1] * w[2] - u[2] * w[1], cy = u[2] * w[0] - u[0] * w[2], cz = u[0] * w[1] - u[1] * w[0];
            const double a2 = cx * cx + cy * cy + cz * cz;               // (2 area)^2
            if (a2 > 4e-16) { const double R2 = la2 * lb2 * lc2 * 0.25 * stat_rcp(a2); if (R2 < min_r2) min_r2 = R2; }
        }
    }
}
constexpr double STAT_NO_RADIUS2 = 1.0e24;   // (1e12)^2: the reference's initial min_r
__device__ __forceinline__ double stat_climb(double max_climb2) { return sqrt(max_climb2); }
__device__ __forceinline__ double stat_radius(double min_r2) { return min_r2 >= STAT_NO_RADIUS2 ? 1.0e12 : sqrt(min_r2); }

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
    double max_climb = 0.0, min_r = STAT_NO_RADIUS2;   // squares, see stat_look
    auto record = [&](const double (&p)[3]) {
        if (n < a.capacity) { out[n * 3] = (IO)p[0]; out[n * 3 + 1] = (IO)p[1]; out[n * 3 + 2] = (IO)p[2]; }
        stat_look(p0, p1, p, n >= 1, n >= 2, max_climb, min_r);   // statistics over consecutive recorded samples (:167-193), squared
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
        for (double t = dt; t <= t_end(T); t += dt) {  // accumulated like the reference (:140)
            eval_poly<M>(c, t < T ? t : T, cur);
            const double dx = cur[0] - prev[0], dy = cur[1] - prev[1], dz = cur[2] - prev[2];
            if (dx * dx + dy * dy + dz * dz >= a.keep_dist2) {   // <=> sqrt(.) >= sample_distance, see SampleArgs
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
    if (a.stats) { a.stats[b * 2] = stat_climb(max_climb); a.stats[b * 2 + 1] = stat_radius(min_r); }
}

// ---------------------------------------------------------------------------------------------
// Segment-parallel sampler (trajectories of <= 64 segments): one lane per (trajectory, segment).
// What makes that legal: the reference restarts its thinning reference point at every segment
// (`prev = eval(0)`, :128-133), so which candidates a segment keeps does not depend on the other
// segments.  What does cross segments is (a) the output position = number of samples recorded
// before, (b) the two most recently recorded samples, which the climb-rate / turn-radius statistics
// of a segment's first samples and the end-point rule look at.  So:
//   pass 1  every lane evaluates and thins its segment without storing: count, last two kept samples;
//   exchange: prefix sum of the counts over the trajectory's lanes, the two most recent samples
//           recorded before each segment (backward scan), the end-point decision of the last segment;
//   pass 2  the same evaluation and thinning again, now storing at the known offset.
// The statistics (:167-193) are a function of consecutive recorded samples only.  With fp64 storage
// the stored samples ARE the recorded values, so a second kernel (sample_stats_kernel: one wave per
// trajectory, coalesced reads, max/min reductions) computes them from the sample array; a trajectory
// that overflows `capacity` (samples beyond it are not stored) and fp32 storage (stored values are
// rounded) take them inline in pass 2 instead, like the one-lane kernel.  The arithmetic per
// candidate and per recorded sample is the one-lane kernel's: identical samples, counts, statistics.
struct SampleStats {
    double p0[3], p1[3];   // the two most recently recorded samples
    double max_climb, min_r;
    int64_t n;
    // statistics of the sample about to become number n (:167-193); max_climb / min_r hold the SQUARES (stat_look)
    __device__ __forceinline__ void look(const double (&p)[3]) { stat_look(p0, p1, p, n >= 1, n >= 2, max_climb, min_r); }
    __device__ __forceinline__ void shift(const double (&p)[3]) {
        p0[0] = p1[0]; p0[1] = p1[1]; p0[2] = p1[2];
        p1[0] = p[0]; p1[1] = p[1]; p1[2] = p[2];
        ++n;
    }
};

template <int O, typename IO>
__global__ void __launch_bounds__(64) sample_seg_kernel(SampleArgs a, int lpt_log2) {
    constexpr int M = 2 * O;
    constexpr bool F64 = sizeof(IO) == 8;
    __shared__ double l_last[6 * 64];
    __shared__ int l_cnt[64];
    const int lane = threadIdx.x;
    const int lpt = 1 << lpt_log2;
    const int j = lane & (lpt - 1);
    const int base = lane - j;
    const int64_t b = ((int64_t)blockIdx.x * 64 + lane) >> lpt_log2;
    const bool traj_ok = b < a.B;
    const int64_t bb = traj_ok ? b : a.B - 1;
    int64_t seg0;
    int S;
    if (a.seg_off) { seg0 = a.seg_off[bb]; S = (int)(a.seg_off[bb + 1] - seg0); }
    else { seg0 = bb * (int64_t)a.S; S = a.S; }
    if (!traj_ok) S = 0;
    const bool active = j < S;
    const int sj = active ? j : 0;
    const IO *rec;
    if (a.seg_major && !a.seg_off) rec = (const IO *)a.coeffs + ((int64_t)sj * a.B + bb) * 3 * M;
    else rec = (const IO *)a.coeffs + (seg0 + sj) * 3 * M;
    double c[3][M];
    double T = 1.0;
    if (active) {
#pragma unroll
        for (int ax = 0; ax < 3; ++ax)
#pragma unroll
            for (int k = 0; k < M; ++k) c[ax][k] = (double)rec[ax * M + k];
        T = (double)((const IO *)a.times)[seg0 + sj];
    } else {
#pragma unroll
        for (int ax = 0; ax < 3; ++ax)
#pragma unroll
            for (int k = 0; k < M; ++k) c[ax][k] = 0.0;
    }
    double dt = 0.1;
    if (dt > T / 10.0) dt = T / 10.0;   // at least 10 evaluations per segment (:126)
    const double sd2 = a.keep_dist2;   // d2 >= sd2  <=>  sqrt(d2) >= sample_distance (SampleArgs)

    // ---- pass 1: which candidates does this segment keep? ----
    int cnt = 0;
    double l1[3] = {0, 0, 0}, l2[3] = {0, 0, 0};   // last / second-to-last kept sample of this segment
    if (active) {
        double prev[3], cur[3];
        eval_poly<M>(c, 0.0, prev);
        if (j == 0) { cnt = 1; l1[0] = prev[0]; l1[1] = prev[1]; l1[2] = prev[2]; }   // the very first sample (n == 0)
        for (double t = dt; t <= t_end(T); t += dt) {
            eval_poly<M>(c, t < T ? t : T, cur);
            const double dx = cur[0] - prev[0], dy = cur[1] - prev[1], dz = cur[2] - prev[2];
            if (dx * dx + dy * dy + dz * dz >= sd2) {
#pragma unroll
                for (int q = 0; q < 3; ++q) { prev[q] = cur[q]; l2[q] = l1[q]; l1[q] = cur[q]; }
                ++cnt;
            }
        }
    }
    l_cnt[lane] = cnt;
#pragma unroll
    for (int q = 0; q < 3; ++q) { l_last[q * 64 + lane] = l1[q]; l_last[(3 + q) * 64 + lane] = l2[q]; }
    __syncthreads();

    // ---- exchange: samples recorded before my segment, and the two most recent of them ----
    int nbefore;
    {
        int incl = cnt;   // inclusive prefix sum over the trajectory's lanes
        for (int d = 1; d < lpt; d <<= 1) {
            const int o = __shfl_up(incl, d, lpt);
            if (j >= d) incl += o;
        }
        nbefore = incl - cnt;
    }
    SampleStats st;
    st.max_climb = 0.0;
    st.min_r = STAT_NO_RADIUS2;
    st.n = nbefore;
#pragma unroll
    for (int q = 0; q < 3; ++q) { st.p0[q] = 0.0; st.p1[q] = 0.0; }
    {
        int have = 0;
        for (int i = j - 1; i >= 0 && have < 2; --i) {
            const int ci = l_cnt[base + i];
            if (ci >= 1) {
                if (have == 0) {
#pragma unroll
                    for (int q = 0; q < 3; ++q) st.p1[q] = l_last[q * 64 + base + i];
                    have = 1;
                    if (ci >= 2) {
#pragma unroll
                        for (int q = 0; q < 3; ++q) st.p0[q] = l_last[(3 + q) * 64 + base + i];
                        have = 2;
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < 3; ++q) st.p0[q] = l_last[q * 64 + base + i];
                    have = 2;
                }
            }
        }
    }
    // end-point rule of the last segment (:157-160): present unless it duplicates the last recorded sample
    bool add_end = false;
    double pend[3] = {0, 0, 0};
    if (active && j == S - 1) {
        eval_poly<M>(c, T, pend);
        const double *pl = cnt >= 1 ? l1 : st.p1;
        const double dx = pl[0] - pend[0], dy = pl[1] - pend[1], dz = pl[2] - pend[2];
        add_end = (nbefore + cnt == 0) || sqrt(dx * dx + dy * dy + dz * dz) > 1e-6;
    }
    // the trajectory's total, known to its last lane, to every lane: does it fit `capacity`?
    __syncthreads();
    if (active && j == S - 1) l_cnt[base] = nbefore + cnt + (add_end ? 1 : 0);
    __syncthreads();
    const int traj_total = S > 0 ? l_cnt[base] : 0;
    // With the squared statistics (stat_look: ~40 instructions per recorded sample) the second pass takes them inline for
    // every trajectory; the separate statistics kernel (one wave per trajectory re-reading the samples: two dependent memory
    // round trips per wave, 70-90 us at B = 65536 whatever its arithmetic) is gone.
    const bool inline_stats = a.stats != nullptr;
    (void)F64; (void)traj_total;

    // ---- pass 2: evaluate and thin again, now storing ----
    IO *out = (IO *)a.samples + bb * a.capacity * 3;
    auto record = [&](const double (&p)[3]) {
        if (st.n < a.capacity) { out[st.n * 3] = (IO)p[0]; out[st.n * 3 + 1] = (IO)p[1]; out[st.n * 3 + 2] = (IO)p[2]; }
        if (inline_stats) st.look(p);
        st.shift(p);
    };
    if (active) {
        double prev[3], cur[3];
        eval_poly<M>(c, 0.0, prev);
        if (j == 0) record(prev);
        for (double t = dt; t <= t_end(T); t += dt) {
            eval_poly<M>(c, t < T ? t : T, cur);
            const double dx = cur[0] - prev[0], dy = cur[1] - prev[1], dz = cur[2] - prev[2];
            if (dx * dx + dy * dy + dz * dz >= sd2) {
                prev[0] = cur[0]; prev[1] = cur[1]; prev[2] = cur[2];
                record(cur);
            }
        }
        if (add_end) record(pend);
    }
    // ---- per-trajectory results: max / min over the lanes ----
    double max_climb = st.max_climb, min_r = st.min_r;
    for (int d = 1; d < lpt; d <<= 1) {
        const double oc = __shfl_xor(max_climb, d, 64), orr = __shfl_xor(min_r, d, 64);
        max_climb = oc > max_climb ? oc : max_climb;
        min_r = orr < min_r ? orr : min_r;
    }
    if (traj_ok && (S == 0 ? j == 0 : j == S - 1)) {
        a.counts[b] = (int32_t)st.n;
        if (a.stats) { a.stats[b * 2] = stat_climb(max_climb); a.stats[b * 2 + 1] = stat_radius(min_r); }
    }
}

// Statistics from the stored samples (fp64 storage, trajectories that fit `capacity`): one wave per
// trajectory, lane i looks at samples i-2, i-1, i.
__global__ void __launch_bounds__(256) sample_stats_kernel(const double *samples, const int32_t *counts, double *stats,
                                                           int64_t B, int64_t capacity) {
    const int lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const int n = counts[b];
    if (n > capacity) return;   // taken inline by sample_seg_kernel
    const double *row = samples + b * capacity * 3;
    SampleStats st;
    st.max_climb = 0.0;
    st.min_r = STAT_NO_RADIUS2;
    for (int i = lane; i < n; i += 64) {
        double p[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            p[q] = row[i * 3 + q];
            st.p1[q] = i >= 1 ? row[(i - 1) * 3 + q] : 0.0;
            st.p0[q] = i >= 2 ? row[(i - 2) * 3 + q] : 0.0;
        }
        st.n = i;
        st.look(p);
    }
    double max_climb = st.max_climb, min_r = st.min_r;
    for (int d = 1; d < 64; d <<= 1) {
        const double oc = __shfl_xor(max_climb, d, 64), orr = __shfl_xor(min_r, d, 64);
        max_climb = oc > max_climb ? oc : max_climb;
        min_r = orr < min_r ? orr : min_r;
    }
    if (lane == 0) { stats[b * 2] = stat_climb(max_climb); stats[b * 2 + 1] = stat_radius(min_r); }
}

// Candidate times are ACCUMULATED (t += dt, minimum_snap.cpp:136) and their rounding decides which candidates exist
// and where they lie, so a wave that wants 64 of them at once has to run the 63 dependent additions every round -- a
// third of this kernel's time for a flight of six 60-second legs.  With dt = 0.1 (every segment of >= 1 s) the sequence
// is the same for every segment of every call: `tacc[k]` = the (k+1)-th accumulated time, built once per device by one
// lane doing the additions in order (tacc_init_kernel), turns the 63 additions into one coalesced load.
// A window switches to the "dense" chain (every lane finds its own successor among the next 16 candidates through LDS,
// ~2 k clocks whatever the window holds) when the previous window recorded at least this many samples; the sparse chain
// costs one ballot round (~130 clocks) per RECORDED sample.  Measured on one flight of six 12-km legs (3600 candidates):
// threshold 4 / 8 / 16 / never: 74.6 / 64.6 / 64.8 / 64.6 us at 166 samples, 112 / 109 / 112 / 139 us at 1149 samples.
constexpr int CSP_DENSE_MIN = 16;
constexpr int TACC_N = 8192;   // candidates per segment covered by the table (819 s of flight per segment)
__global__ void tacc_init_kernel(double *tab, int n) {
    double t = 0.1;
    for (int k = 0; k < n; ++k) { tab[k] = t; t += 0.1; }
}

// ---------------------------------------------------------------------------------------------
// Wave-cooperative sampler for LONG segments (hundreds to thousands of candidates each: kilometre
// legs sampled every 0.1 s, the reference's own use -- one flight per call).  One wave per
// trajectory, segments in order, 64 candidates per round:
//   * candidate times are the reference's running sum (`t += dt`, :140), not k*dt -- after thousands
//     of additions the two differ by more than the loop's 1e-12 slack -- so a window's 64 times come
//     from 64 sequential additions captured lane by lane;
//   * all 64 candidates are evaluated at once; the thinning inside the window is a chain: the first
//     candidate at distance >= sample_distance from `prev`, then from each recorded candidate the
//     first later one far enough from IT.  Sparse windows follow the chain with one ballot +
//     count-trailing-zeros per recorded sample; windows that record many samples first let every lane
//     find its own successor among the next 16 candidates (LDS), after which the chain advances by
//     register reads.  The whole chain is then stored at once (rank in the chain = output row);
//   * the climb-rate / turn-radius statistics (:167-193) need (sample i-2, i-1, i) only: recorded
//     samples queue up in an LDS ring and every 64 of them are processed by the 64 lanes at once.
// Same power sums, squared-distance test and statistics formulas as the other two samplers:
// identical samples, counts and statistics.
template <int O, typename IO>
__global__ void __launch_bounds__(64) sample_wave_kernel(SampleArgs a, const double *tacc) {
    constexpr int M = 2 * O;
    __shared__ double ring[130 * 3];  // [0],[1]: the two samples before the queued batch; [2..129]: the batch (<= 127)
    __shared__ double lc[3 * 64];     // the window's candidates, for the successor look-ahead
    const int lane = threadIdx.x;
    const int64_t b = blockIdx.x;
    int64_t seg0;
    int S;
    if (a.seg_off) { seg0 = a.seg_off[b]; S = (int)(a.seg_off[b + 1] - seg0); }
    else { seg0 = b * (int64_t)a.S; S = a.S; }
    IO *out = (IO *)a.samples + b * a.capacity * 3;
    SampleStats st;          // per-lane partial max / min; p0, p1, n are filled per sample in flush()
    st.max_climb = 0.0;
    st.min_r = STAT_NO_RADIUS2;
    int64_t n = 0;           // samples recorded so far (wave-uniform)
    int nb = 0;              // of which queued in the ring
    double last[3] = {0, 0, 0};   // the most recently recorded sample (wave-uniform)
    auto flush = [&]() {
        __syncthreads();
        for (int e = lane; e < nb; e += 64) {
            double p[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) { p[q] = ring[(2 + e) * 3 + q]; st.p1[q] = ring[(1 + e) * 3 + q]; st.p0[q] = ring[e * 3 + q]; }
            st.n = n - nb + e;
            st.look(p);
        }
        __syncthreads();
        if (lane < 6) ring[lane] = ring[nb * 3 + lane];   // the last two become the predecessors of the next batch
        nb = 0;
        __syncthreads();
    };
    auto record = [&](const double (&p)[3]) {   // wave-uniform argument
        if (lane < 3) {
            if (n < a.capacity) out[n * 3 + lane] = (IO)p[lane == 0 ? 0 : lane == 1 ? 1 : 2];
            ring[(2 + nb) * 3 + lane] = p[lane == 0 ? 0 : lane == 1 ? 1 : 2];
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) last[q] = p[q];
        ++n;
        ++nb;
        if (nb >= 64) flush();
    };
    auto bcast = [](double v, int src) {
        const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
        const unsigned lo = __builtin_amdgcn_readlane((unsigned)u, src), hi = __builtin_amdgcn_readlane((unsigned)(u >> 32), src);
        return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
    };
    const double sd2 = a.keep_dist2;
    bool dense = false;      // the previous window recorded >= CSP_DENSE_MIN samples (wave-uniform)
    for (int seg = 0; seg < S; ++seg) {
        const IO *rec = (a.seg_major && !a.seg_off) ? (const IO *)a.coeffs + ((int64_t)seg * a.B + b) * 3 * M
                                                     : (const IO *)a.coeffs + (seg0 + seg) * 3 * M;
        double c[3][M];
#pragma unroll
        for (int ax = 0; ax < 3; ++ax)
#pragma unroll
            for (int k = 0; k < M; ++k) c[ax][k] = (double)rec[ax * M + k];
        const double T = (double)((const IO *)a.times)[seg0 + seg];
        double dt = 0.1;
        if (dt > T / 10.0) dt = T / 10.0;
        double prev[3];
        eval_poly<M>(c, 0.0, prev);
        if (n == 0) record(prev);
        double tb = dt;   // accumulated time of the window's first candidate
        const bool use_tab = tacc != nullptr && dt == 0.1;
        int k0 = 0;       // index of tb in the accumulated sequence
        double t_pref = use_tab ? tacc[lane] : 0.0;
        while (tb <= t_end(T)) {
            // the window's 64 candidate times: lane L performs the first L of the 63 sequential additions
            // (the same partial sums, in the same order, as the reference's running t) -- or reads them from the
            // table of those sums when dt = 0.1
            double t = tb;
            if (use_tab && k0 + 64 <= TACC_N) {
                t = t_pref;
            } else {
                for (int i = 0; i < 63; ++i) {
                    if (lane > i) t += dt;
                }
            }
            k0 += 64;
            if (use_tab && k0 + 64 <= TACC_N) t_pref = tacc[k0 + lane];
            const double run = bcast(t, 63) + dt;   // the accumulated time of the next window's first candidate
            const bool exists = t <= t_end(T);
            double cur[3];
            eval_poly<M>(c, t < T ? t : T, cur);
            double dx = cur[0] - prev[0], dy = cur[1] - prev[1], dz = cur[2] - prev[2];
            const unsigned long long keep = __builtin_amdgcn_ballot_w64(exists && (dx * dx + dy * dy + dz * dz >= sd2));
            if (keep != 0) {
                // The recorded samples of this window form a chain: the first candidate far enough from `prev`,
                // then from each recorded candidate l the first later one far enough from IT.
                int l = __builtin_ctzll(keep);
                unsigned long long kept = 0;
                int nxt = 64;   // dense mode: my successor if it lies within the next LA lanes
                constexpr int LA = 16;
                if (dense) {
                    // many samples per window: every lane finds its own successor among the next LA
                    // candidates at once (through LDS), so the chain below advances by register reads
                    __syncthreads();
#pragma unroll
                    for (int q = 0; q < 3; ++q) lc[q * 64 + lane] = cur[q];
                    __syncthreads();
                    const int E = __builtin_popcountll(__builtin_amdgcn_ballot_w64(exists));   // existing candidates are lanes 0..E-1
                    for (int s2 = 1; s2 <= LA; ++s2) {
                        const int m = lane + s2;
                        if (m < E && nxt == 64) {
                            const double ex = lc[m] - cur[0], ey = lc[64 + m] - cur[1], ez = lc[128 + m] - cur[2];
                            if (ex * ex + ey * ey + ez * ez >= sd2) nxt = m;
                        }
                    }
                }
                for (;;) {
                    l = __builtin_amdgcn_readfirstlane(l);
                    kept |= 1ull << l;
                    if (dense) {
                        const int nx = __builtin_amdgcn_readlane(nxt, l);
                        if (nx < 64) { l = nx; continue; }
                    }
                    // ballot search from candidate l (in dense mode: beyond its look-ahead)
                    double pl[3];
#pragma unroll
                    for (int q = 0; q < 3; ++q) pl[q] = bcast(cur[q], l);
                    dx = cur[0] - pl[0]; dy = cur[1] - pl[1]; dz = cur[2] - pl[2];
                    const unsigned long long more = __builtin_amdgcn_ballot_w64(exists && lane > l + (dense ? LA : 0) &&
                                                                               (dx * dx + dy * dy + dz * dz >= sd2));
                    if (more == 0) break;
                    l = __builtin_ctzll(more);
                }
                // record the whole chain at once: recorded lane -> output row n + (its rank in the chain)
                const int cnt = __builtin_popcountll(kept);
                if ((kept >> lane) & 1ull) {
                    const int rank = __builtin_popcountll(kept & ((1ull << lane) - 1ull));
                    if (n + rank < a.capacity) {
#pragma unroll
                        for (int q = 0; q < 3; ++q) out[(n + rank) * 3 + q] = (IO)cur[q];
                    }
#pragma unroll
                    for (int q = 0; q < 3; ++q) ring[(2 + nb + rank) * 3 + q] = cur[q];
                }
                const int lastl = 63 - __builtin_clzll(kept);
#pragma unroll
                for (int q = 0; q < 3; ++q) { prev[q] = bcast(cur[q], lastl); last[q] = prev[q]; }
                n += cnt;
                nb += cnt;
                if (nb >= 64) flush();
                dense = cnt >= CSP_DENSE_MIN;
            } else {
                dense = false;
            }
            tb = run;   // = the accumulated time of candidate 64 of this window
        }
        if (seg == S - 1) {  // end-point rule (:157-160)
            double cur[3];
            eval_poly<M>(c, T, cur);
            const double ex = last[0] - cur[0], ey = last[1] - cur[1], ez = last[2] - cur[2];
            if (n == 0 || sqrt(ex * ex + ey * ey + ez * ez) > 1e-6) record(cur);
        }
    }
    flush();
    double max_climb = st.max_climb, min_r = st.min_r;
    for (int d = 1; d < 64; d <<= 1) {
        const double oc = __shfl_xor(max_climb, d, 64), orr = __shfl_xor(min_r, d, 64);
        max_climb = oc > max_climb ? oc : max_climb;
        min_r = orr < min_r ? orr : min_r;
    }
    if (lane == 0) {
        a.counts[b] = (int32_t)n;
        if (a.stats) { a.stats[b * 2] = stat_climb(max_climb); a.stats[b * 2 + 1] = stat_radius(min_r); }
    }
}

// ---------------------------------------------------------------------------------------------
// One wave per SEGMENT (a few long flights from host memory -- the reference's own call: one flight, six kilometre
// legs, 3600 candidates): the candidate search of sample_wave_kernel for one segment, recording into the segment's own
// run of `tmp`.  Nothing crosses segments here: the first-sample rule concerns a trajectory's first segment only, the
// end-point rule and the statistics are applied after the runs have been placed (sample_place_kernel,
// sample_stats_kernel).  Same accumulated candidate times, same squared-distance test, same chain: the recorded
// samples are bitwise those of the other samplers.
template <int O>
__global__ void __launch_bounds__(64) sample_wave_seg_kernel(SampleArgs a, double *tmp, const int64_t *tmp_off, int32_t *seg_counts,
                                                             const double *tacc) {
    constexpr int M = 2 * O;
    __shared__ double lc[3 * 64];
    const int lane = threadIdx.x;
    const int64_t g = blockIdx.x;                 // global segment
    int64_t b, seg0;
    int S;
    if (a.seg_off) {                              // few trajectories: a linear search is fine
        b = 0;
        while (b + 1 < a.B && a.seg_off[b + 1] <= g) ++b;
        seg0 = a.seg_off[b];
        S = (int)(a.seg_off[b + 1] - seg0);
    } else { b = g / a.S; seg0 = b * (int64_t)a.S; S = a.S; }
    const int seg = (int)(g - seg0);
    double *out = tmp + tmp_off[g] * 3;
    const int64_t room = tmp_off[g + 1] - tmp_off[g];
    const double *rec = (const double *)a.coeffs + g * 3 * M;
    double c[3][M];
#pragma unroll
    for (int ax = 0; ax < 3; ++ax)
#pragma unroll
        for (int k = 0; k < M; ++k) c[ax][k] = rec[ax * M + k];
    const double T = ((const double *)a.times)[g];
    double dt = 0.1;
    if (dt > T / 10.0) dt = T / 10.0;
    auto bcast = [](double v, int src) {
        const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
        const unsigned lo = __builtin_amdgcn_readlane((unsigned)u, src), hi = __builtin_amdgcn_readlane((unsigned)(u >> 32), src);
        return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
    };
    const double sd2 = a.keep_dist2;
    int64_t n = 0;
    double prev[3];
    eval_poly<M>(c, 0.0, prev);
    if (seg == 0) {                               // the trajectory's very first sample (:128-133, n == 0)
        if (lane < 3 && n < room) out[n * 3 + lane] = prev[lane == 0 ? 0 : lane == 1 ? 1 : 2];
        ++n;
    }
    bool dense = false;
    double tb = dt;
    const bool use_tab = tacc != nullptr && dt == 0.1;
    int k0 = 0;                                   // index of tb in the accumulated sequence
    double t_pref = use_tab ? tacc[lane] : 0.0;
    while (tb <= t_end(T)) {
        double t = tb;
        if (use_tab && k0 + 64 <= TACC_N) {
            t = t_pref;
        } else {
            for (int i = 0; i < 63; ++i) {
                if (lane > i) t += dt;
            }
        }
        k0 += 64;
        if (use_tab && k0 + 64 <= TACC_N) t_pref = tacc[k0 + lane];   // the next round's times, in flight during this round
        const double run = bcast(t, 63) + dt;
        const bool exists = t <= t_end(T);
        double cur[3];
        eval_poly<M>(c, t < T ? t : T, cur);
        double dx = cur[0] - prev[0], dy = cur[1] - prev[1], dz = cur[2] - prev[2];
        const unsigned long long keep = __builtin_amdgcn_ballot_w64(exists && (dx * dx + dy * dy + dz * dz >= sd2));
        if (keep != 0) {
            int l = __builtin_ctzll(keep);
            unsigned long long kept = 0;
            int nxt = 64;
            constexpr int LA = 16;
            if (dense) {
                __syncthreads();
#pragma unroll
                for (int q = 0; q < 3; ++q) lc[q * 64 + lane] = cur[q];
                __syncthreads();
                const int E = __builtin_popcountll(__builtin_amdgcn_ballot_w64(exists));
                for (int s2 = 1; s2 <= LA; ++s2) {
                    const int m = lane + s2;
                    if (m < E && nxt == 64) {
                        const double ex = lc[m] - cur[0], ey = lc[64 + m] - cur[1], ez = lc[128 + m] - cur[2];
                        if (ex * ex + ey * ey + ez * ez >= sd2) nxt = m;
                    }
                }
            }
            for (;;) {
                l = __builtin_amdgcn_readfirstlane(l);
                kept |= 1ull << l;
                if (dense) {
                    const int nx = __builtin_amdgcn_readlane(nxt, l);
                    if (nx < 64) { l = nx; continue; }
                }
                double pl[3];
#pragma unroll
                for (int q = 0; q < 3; ++q) pl[q] = bcast(cur[q], l);
                dx = cur[0] - pl[0]; dy = cur[1] - pl[1]; dz = cur[2] - pl[2];
                const unsigned long long more = __builtin_amdgcn_ballot_w64(exists && lane > l + (dense ? LA : 0) &&
                                                                           (dx * dx + dy * dy + dz * dz >= sd2));
                if (more == 0) break;
                l = __builtin_ctzll(more);
            }
            const int cnt = __builtin_popcountll(kept);
            if ((kept >> lane) & 1ull) {
                const int rank = __builtin_popcountll(kept & ((1ull << lane) - 1ull));
                if (n + rank < room) {
#pragma unroll
                    for (int q = 0; q < 3; ++q) out[(n + rank) * 3 + q] = cur[q];
                }
            }
            const int lastl = 63 - __builtin_clzll(kept);
#pragma unroll
            for (int q = 0; q < 3; ++q) prev[q] = bcast(cur[q], lastl);
            n += cnt;
            dense = cnt >= CSP_DENSE_MIN;
        } else {
            dense = false;
        }
        tb = run;
    }
    if (seg == S - 1) {   // the end point p(T), for the end-point rule: parked in the run's LAST slot (never reached by samples)
        double pe[3];
        eval_poly<M>(c, T, pe);
        if (lane < 3) out[(room - 1) * 3 + lane] = pe[lane == 0 ? 0 : lane == 1 ? 1 : 2];
    }
    if (lane == 0) seg_counts[g] = (int32_t)n;
}

// Places the per-segment runs of sample_wave_seg_kernel: one workgroup per trajectory; prefix sum of the segment counts, coalesced
// copies, the end-point rule (:157-160: present unless it duplicates the last recorded sample), the count, the statistics.
// FOUR waves per trajectory and four elements in flight per lane: with one wave and one element the copy is a chain of
// dependent load -> store round trips (47 us for 2129 samples).
__global__ void __launch_bounds__(256) sample_place_kernel(SampleArgs a, const double *tmp, const int64_t *tmp_off, const int32_t *seg_counts,
                                                           LoopUpdate upd) {
    __shared__ double l_red[2 * 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t b = blockIdx.x;
    int64_t seg0;
    int S;
    if (a.seg_off) { seg0 = a.seg_off[b]; S = (int)(a.seg_off[b + 1] - seg0); }
    else { seg0 = b * (int64_t)a.S; S = a.S; }
    double *out = (double *)a.samples + b * a.capacity * 3;
    const int64_t room = a.capacity * 3;
    int64_t n = 0;
    for (int seg = 0; seg < S; ++seg) {
        const int64_t g = seg0 + seg;
        const int64_t cnt3 = (int64_t)seg_counts[g] * 3;
        const double *src = tmp + tmp_off[g] * 3;
        for (int64_t e0 = tid; e0 < cnt3; e0 += 4 * 256) {
            double v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t e = e0 + u * 256;
                v[u] = src[e < cnt3 ? e : cnt3 - 1];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t e = e0 + u * 256;
                if (e < cnt3 && n * 3 + e < room) out[n * 3 + e] = v[u];
            }
        }
        n += seg_counts[g];
    }
    if (S > 0) {
        const int64_t gl = seg0 + S - 1;
        const double *pe = tmp + (tmp_off[gl + 1] - 1) * 3;
        bool add = n == 0;
        if (n > 0) {
            // the last recorded sample: the last one of the last non-empty run
            int sl = S - 1;
            while (sl > 0 && seg_counts[seg0 + sl] == 0) --sl;
            const double *lp = tmp + (tmp_off[seg0 + sl] + seg_counts[seg0 + sl] - 1) * 3;
            const double dx = lp[0] - pe[0], dy = lp[1] - pe[1], dz = lp[2] - pe[2];
            add = sqrt(dx * dx + dy * dy + dz * dz) > 1e-6;
        }
        if (add) {
            if (tid < 3 && n < a.capacity) out[n * 3 + tid] = pe[tid];
            ++n;
        }
    }
    if (tid == 0) a.counts[b] = (int32_t)n;
    if (tid == 0 && upd.done && !upd.done[b]) {   // resolve_update_kernel's step for this trajectory (first pass of the loop)
        if (upd.max_dev[b] > 0.2 && upd.iters[b] < 10) {
            const double w = upd.vw[b];
            upd.vw[b] = (w < 1e-6) ? 0.01 : w * 2.0;
            upd.iters[b] += 1;
            if (upd.pending) atomicAdd(upd.pending, 1);
        } else {
            upd.done[b] = 1;
        }
    }
    if (a.stats) {
        // the statistics (:167-193) from the rows this workgroup has just placed (n <= capacity here): its stores are
        // complete and visible after fence + barrier; same per-sample code as sample_stats_kernel, which a separate
        // launch would cost ~5 us for
        __threadfence();
        __syncthreads();
        SampleStats st;
        st.max_climb = 0.0;
        st.min_r = STAT_NO_RADIUS2;
        for (int64_t i = tid; i < n; i += 256) {
            double p[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                p[q] = out[i * 3 + q];
                st.p1[q] = i >= 1 ? out[(i - 1) * 3 + q] : 0.0;
                st.p0[q] = i >= 2 ? out[(i - 2) * 3 + q] : 0.0;
            }
            st.n = i;
            st.look(p);
        }
        double max_climb = st.max_climb, min_r = st.min_r;
        for (int d = 1; d < 64; d <<= 1) {
            const double oc = __shfl_xor(max_climb, d, 64), orr = __shfl_xor(min_r, d, 64);
            max_climb = oc > max_climb ? oc : max_climb;
            min_r = orr < min_r ? orr : min_r;
        }
        if (lane == 0) { l_red[wave] = max_climb; l_red[4 + wave] = min_r; }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < 4; ++w) {
                max_climb = l_red[w] > max_climb ? l_red[w] : max_climb;
                min_r = l_red[4 + w] < min_r ? l_red[4 + w] : min_r;
            }
            a.stats[b * 2] = stat_climb(max_climb);
            a.stats[b * 2 + 1] = stat_radius(min_r);
        }
    }
}

// The per-device table of accumulated candidate times (64 KB, built at the first call, kept for the life of the process).
static const double *tacc_table(hipStream_t st) {
    struct Slot { std::once_flag once; double *tab = nullptr; };
    static Slot slots[64];
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    Slot &sl = slots[dev];
    std::call_once(sl.once, [&] {
        double *p = nullptr;
        if (hipMalloc((void **)&p, sizeof(double) * TACC_N) != hipSuccess) { (void)hipGetLastError(); return; }
        hipLaunchKernelGGL(tacc_init_kernel, dim3(1), dim3(1), 0, st, p, TACC_N);
        if (hipGetLastError() == hipSuccess && hipStreamSynchronize(st) == hipSuccess) sl.tab = p;
        else (void)hipFree(p);
    });
    return sl.tab;   // null = the kernels accumulate by themselves
}

hipError_t launch_sample_segment_waves(const SampleArgs &a, double *tmp, const int64_t *tmp_off, int32_t *seg_counts,
                                       int64_t total_segments, hipStream_t st) {
    return launch_sample_segment_waves_upd(a, tmp, tmp_off, seg_counts, total_segments, nullptr, st);
}

hipError_t launch_sample_segment_waves_upd(const SampleArgs &a, double *tmp, const int64_t *tmp_off, int32_t *seg_counts,
                                           int64_t total_segments, const LoopUpdate *upd, hipStream_t st) {
    if (a.B == 0 || total_segments == 0) return hipSuccess;
    const dim3 grid((unsigned)total_segments), block(64);
    const double *tacc = tacc_table(st);
    switch (a.order) {
        case 1: hipLaunchKernelGGL((sample_wave_seg_kernel<1>), grid, block, 0, st, a, tmp, tmp_off, seg_counts, tacc); break;
        case 2: hipLaunchKernelGGL((sample_wave_seg_kernel<2>), grid, block, 0, st, a, tmp, tmp_off, seg_counts, tacc); break;
        case 3: hipLaunchKernelGGL((sample_wave_seg_kernel<3>), grid, block, 0, st, a, tmp, tmp_off, seg_counts, tacc); break;
        case 4: hipLaunchKernelGGL((sample_wave_seg_kernel<4>), grid, block, 0, st, a, tmp, tmp_off, seg_counts, tacc); break;
        case 5: hipLaunchKernelGGL((sample_wave_seg_kernel<5>), grid, block, 0, st, a, tmp, tmp_off, seg_counts, tacc); break;
        default: return hipErrorInvalidValue;
    }
    const LoopUpdate none = {nullptr, nullptr, nullptr, nullptr, nullptr};
    hipLaunchKernelGGL(sample_place_kernel, dim3((unsigned)a.B), dim3(256), 0, st, a, (const double *)tmp, tmp_off, (const int32_t *)seg_counts,
                       upd ? *upd : none);
    return hipGetLastError();
}

template <typename IO> static hipError_t launch_sample_t(const SampleArgs &a, hipStream_t st) {
    if (a.long_segments && !a.one_lane) {   // one wave per trajectory, 64 candidates per round
        const dim3 grid((unsigned)a.B), block(64);
        const double *tacc = tacc_table(st);
        switch (a.order) {
            case 1: hipLaunchKernelGGL((sample_wave_kernel<1, IO>), grid, block, 0, st, a, tacc); break;
            case 2: hipLaunchKernelGGL((sample_wave_kernel<2, IO>), grid, block, 0, st, a, tacc); break;
            case 3: hipLaunchKernelGGL((sample_wave_kernel<3, IO>), grid, block, 0, st, a, tacc); break;
            case 4: hipLaunchKernelGGL((sample_wave_kernel<4, IO>), grid, block, 0, st, a, tacc); break;
            case 5: hipLaunchKernelGGL((sample_wave_kernel<5, IO>), grid, block, 0, st, a, tacc); break;
            default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
    if (a.Smax >= 1 && a.Smax <= 64 && !a.one_lane) {   // one lane per (trajectory, segment)
        int l = 0;
        while ((1 << l) < a.Smax) ++l;
        const int64_t lanes = a.B << l;
        const dim3 grid((unsigned)((lanes + 63) / 64)), block(64);
        switch (a.order) {
            case 1: hipLaunchKernelGGL((sample_seg_kernel<1, IO>), grid, block, 0, st, a, l); break;
            case 2: hipLaunchKernelGGL((sample_seg_kernel<2, IO>), grid, block, 0, st, a, l); break;
            case 3: hipLaunchKernelGGL((sample_seg_kernel<3, IO>), grid, block, 0, st, a, l); break;
            case 4: hipLaunchKernelGGL((sample_seg_kernel<4, IO>), grid, block, 0, st, a, l); break;
            case 5: hipLaunchKernelGGL((sample_seg_kernel<5, IO>), grid, block, 0, st, a, l); break;
            default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
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
