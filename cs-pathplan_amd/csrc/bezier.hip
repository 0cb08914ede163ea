// bezier.hip -- batched Bezier path sampler (include/csp_bezier.h; reference math_util/bezier.cpp:28-190).
// One WAVE per path, one lane per segment (paths of more than 64 segments take several rounds): a lane derives its
// segment's headings and control points (the <= 10-step curvature loop), counts its samples with the reference's
// accumulated parameter loop, the wave's prefix sum places every segment in the path's sample array, and the lane
// evaluates again and stores.  Per-segment evaluation, no solve: the parallelism is paths x segments.
#include "../../include/csp_bezier.h"
#include "../../include/csp_minsnap.h"
#include "minsnap_hoststage.h"

#include <hip/hip_runtime.h>
#include <cmath>

namespace {

struct BezArgs {
    const double *wp;
    const int64_t *off;
    double *samples;
    int32_t *counts;
    int64_t B, capacity;
    double resolution, min_radius;
};

struct Seg4 { double p[4][3]; };

// control points for arm factor k (bezier.cpp:45-51, :97-103)
__device__ __forceinline__ void place(const double (&a)[3], const double (&d)[3], double ha, double hd, double chord, double k, Seg4 &s) {
    for (int q = 0; q < 3; ++q) { s.p[0][q] = a[q]; s.p[3][q] = d[q]; }
    s.p[1][0] = a[0] + cos(ha) * chord * k;
    s.p[1][1] = a[1] + sin(ha) * chord * k;
    s.p[1][2] = a[2] + (d[2] - a[2]) * 1.0 / 3.0;
    s.p[2][0] = d[0] - cos(hd) * chord * k;
    s.p[2][1] = d[1] - sin(hd) * chord * k;
    s.p[2][2] = a[2] + (d[2] - a[2]) * 2.0 / 3.0;
}

// curvature |v x acc| / |v|^3 at t against 1/min_radius (bezier.cpp:56-85)
__device__ __forceinline__ bool too_tight(const Seg4 &s, double t, double min_radius) {
    const double u = 1.0 - t;
    double v[3], w[3];
    for (int q = 0; q < 3; ++q) {
        v[q] = 3 * u * u * (s.p[1][q] - s.p[0][q]) + 6 * u * t * (s.p[2][q] - s.p[1][q]) + 3 * t * t * (s.p[3][q] - s.p[2][q]);
        w[q] = 6 * u * (s.p[2][q] - 2 * s.p[1][q] + s.p[0][q]) + 6 * t * (s.p[3][q] - 2 * s.p[2][q] + s.p[1][q]);
    }
    const double cx = v[1] * w[2] - v[2] * w[1], cy = v[2] * w[0] - v[0] * w[2], cz = v[0] * w[1] - v[1] * w[0];
    const double speed = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    const double speed3 = speed * speed * speed;
    if (!(speed3 > 1e-6)) return false;
    return sqrt(cx * cx + cy * cy + cz * cz) / speed3 > 1.0 / min_radius;
}

__global__ void __launch_bounds__(64) bezier_kernel(BezArgs a) {
    const int lane = threadIdx.x;
    const int64_t b = blockIdx.x;
    const int64_t p0 = a.off[b];
    const int n = (int)(a.off[b + 1] - p0);      // points of this path
    if (n < 2) { if (lane == 0) a.counts[b] = 0; return; }
    const double *P = a.wp + p0 * 3;
    double *out = a.samples + b * a.capacity * 3;
    int64_t base = 0;                             // samples of the segments already placed (wave-uniform)
    for (int s0 = 0; s0 < n - 1; s0 += 64) {
        const int i = s0 + lane;
        const bool act = i < n - 1;
        const int ii = act ? i : n - 2;
        auto heading = [&](int q) {               // one-sided at the ends, central inside (:144-159)
            const int lo = q == 0 ? 0 : q - 1, hi = q == n - 1 ? n - 1 : q + 1;
            return atan2(P[hi * 3 + 1] - P[lo * 3 + 1], P[hi * 3] - P[lo * 3]);
        };
        double pa[3], pd[3];
        for (int q = 0; q < 3; ++q) { pa[q] = P[ii * 3 + q]; pd[q] = P[(ii + 1) * 3 + q]; }
        const double ha = heading(ii), hd = heading(ii + 1);
        const double chord = hypot(pa[0] - pd[0], pa[1] - pd[1]);
        const bool ok = chord >= 1e-1;            // GeneratePath() == 0 (:37)
        Seg4 sg;
        double k = 1.0 / 3.0;
        for (int attempt = 0; attempt < 10; ++attempt) {
            place(pa, pd, ha, hd, chord, k, sg);
            if (a.min_radius <= 1.0) break;
            if (!(too_tight(sg, 0.0, a.min_radius) || too_tight(sg, 0.5, a.min_radius) || too_tight(sg, 1.0, a.min_radius))) break;
            k += 0.02;
            if (k > 0.45) { k = 0.45; break; }
        }
        place(pa, pd, ha, hd, chord, k, sg);
        const double len = hypot(sg.p[2][0] - sg.p[1][0], sg.p[2][1] - sg.p[1][1]) + chord * 2.0 / 3.0;
        const double step = a.resolution / len;
        // pass 1: how many samples does `for (t = 0; t <= 1; t += step)` produce?  (accumulated, like :110)
        int cnt = 0;
        if (act) {
            if (ok) {
                if (step > 0.0) { for (double t = 0.0; t <= 1.0; t += step) ++cnt; }
                else cnt = 0;                     // a non-positive or NaN step would never end: no samples (not reachable with resolution > 0)
                if (i > 0 && cnt > 0) --cnt;      // segments after the first drop their first sample (:169-171)
            } else {
                cnt = 1;                          // fallback: the end point (:174-178)
            }
        }
        int incl = cnt;
        for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d, 64); if (lane >= d) incl += o; }
        const int64_t my = base + incl - cnt;
        // pass 2: evaluate and store
        if (act) {
            if (ok) {
                int64_t w = my;
                bool skip = i > 0;
                for (double t = 0.0; t <= 1.0 && step > 0.0; t += step) {
                    if (skip) { skip = false; continue; }
                    const double u = 1.0 - t;
                    const double w0 = u * u * u, w1 = 3 * u * u * t, w2 = 3 * u * t * t, w3 = t * t * t;
                    if (w < a.capacity)
                        for (int q = 0; q < 3; ++q) out[w * 3 + q] = w0 * sg.p[0][q] + w1 * sg.p[1][q] + w2 * sg.p[2][q] + w3 * sg.p[3][q];
                    ++w;
                }
            } else if (my < a.capacity) {
                for (int q = 0; q < 3; ++q) out[my * 3 + q] = pd[q];
            }
        }
        base += __shfl(incl, 63, 64);
    }
    if (lane == 0) a.counts[b] = (int32_t)base;
}

}  // namespace

extern "C" int csp_bezier_generate_batch(const double *waypoints, const int64_t *offsets, int64_t batch, double resolution,
                                         double min_radius, int64_t capacity, double *samples, int32_t *counts,
                                         uint32_t mem_space, int32_t device_id, void *hip_stream) {
    if (batch < 0 || capacity < 1 || !(resolution > 0.0) || !(min_radius == min_radius)) return CSP_ERR_INVALID_ARG;
    if (batch == 0) return CSP_OK;
    if (!waypoints || !offsets || !samples || !counts) return CSP_ERR_INVALID_ARG;
    if (mem_space != CSP_MEM_HOST && mem_space != CSP_MEM_DEVICE) return CSP_ERR_INVALID_ARG;
    if (csp_minsnap_device_count() < 1) return CSP_ERR_NO_DEVICE;
    if (device_id >= 0 && hipSetDevice(device_id) != hipSuccess) return CSP_ERR_HIP;
    hipStream_t st = (hipStream_t)hip_stream;
    BezArgs a;
    a.B = batch; a.capacity = capacity; a.resolution = resolution; a.min_radius = min_radius;
    if (mem_space == CSP_MEM_DEVICE) {
        a.wp = waypoints; a.off = offsets; a.samples = samples; a.counts = counts;
        hipLaunchKernelGGL(bezier_kernel, dim3((unsigned)batch), dim3(64), 0, st, a);
        return hipGetLastError() == hipSuccess ? CSP_OK : CSP_ERR_HIP;
    }
    const int64_t total = offsets[batch];
    if (total < 0) return CSP_ERR_INVALID_ARG;
    int cur = 0;
    (void)hipGetDevice(&cur);
    csp::HostCall hc(cur, st);
    const size_t o_wp = hc.in(waypoints, (size_t)total * 24), o_off = hc.in(offsets, (size_t)(batch + 1) * 8);
    const size_t o_ct = hc.out(counts, (size_t)batch * 4), o_sm = hc.out(samples, (size_t)batch * (size_t)capacity * 24);
    if (hc.upload() != hipSuccess) return CSP_ERR_HIP;
    a.wp = hc.ptr<const double>(o_wp); a.off = hc.ptr<const int64_t>(o_off);
    a.samples = hc.ptr<double>(o_sm); a.counts = hc.ptr<int32_t>(o_ct);
    hipLaunchKernelGGL(bezier_kernel, dim3((unsigned)batch), dim3(64), 0, st, a);
    if (hipGetLastError() != hipSuccess || hc.download() != hipSuccess) return CSP_ERR_HIP;
    return CSP_OK;
}
