// minsnap_timealloc.hip -- batched segment-time allocation (GenerateTrajectoryMatrix,
// math_util/minimum_snap.cpp:59-72): T_i = max(|p_{i+1}-p_i| / V_avg, min_time_s).
// One lane per segment; consecutive lanes read consecutive waypoints.
#include "minsnap_timealloc.h"

namespace csp {

template <typename R>
__global__ void __launch_bounds__(256) time_alloc_kernel(TimeAllocArgs a, int64_t total_seg) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total_seg) return;
    int64_t b;
    if (a.seg_off) {
        // ragged: binary search the owning trajectory (seg_off is a prefix sum)
        int64_t lo = 0, hi = a.B;
        while (hi - lo > 1) {
            const int64_t mid = (lo + hi) >> 1;
            if (a.seg_off[mid] <= g) lo = mid; else hi = mid;
        }
        b = lo;
    } else {
        b = g / a.S;
    }
    const R *p = (const R *)a.wp + (g + b) * 3;
    const R dx = p[3] - p[0], dy = p[4] - p[1], dz = p[5] - p[2];
    const R len = sqrt(dx * dx + dy * dy + dz * dz);
    R t = (a.v_avg > 1e-6) ? (R)(len / (R)a.v_avg) : (R)a.min_time_s;
    if (t < (R)a.min_time_s) t = (R)a.min_time_s;
    ((R *)a.times)[g] = t;
}

// Uniform batches: lane g < B*S allocates segment g's time, lane g < B initialises trajectory g's loop state.
template <typename R>
__global__ void __launch_bounds__(256) time_alloc_init_kernel(TimeAllocArgs a, int64_t total_seg, double *vw, int32_t *iters,
                                                              int32_t *done, int32_t *pending, double vw0) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g == 0 && pending) *pending = 0;
    if (g < a.B) { if (vw) vw[g] = vw0; iters[g] = 0; done[g] = 0; }   // vw null: the caller copies per-trajectory weights in
    if (g >= total_seg) return;
    const int64_t b = g / a.S;
    const R *p = (const R *)a.wp + (g + b) * 3;
    const R dx = p[3] - p[0], dy = p[4] - p[1], dz = p[5] - p[2];
    const R len = sqrt(dx * dx + dy * dy + dz * dz);
    R t = (a.v_avg > 1e-6) ? (R)(len / (R)a.v_avg) : (R)a.min_time_s;
    if (t < (R)a.min_time_s) t = (R)a.min_time_s;
    ((R *)a.times)[g] = t;
}

hipError_t launch_time_alloc_init(const TimeAllocArgs &a, bool f32, double *vw, int32_t *iters, int32_t *done, int32_t *pending,
                                  double vw0, hipStream_t st) {
    if (a.seg_off) {   // ragged: the segment total lives on the device (launch_time_alloc fetches it) -- two launches
        hipError_t e = launch_time_alloc(a, f32, st);
        return e != hipSuccess ? e : launch_resolve_init(vw, iters, done, pending, vw0, a.B, st);
    }
    const int64_t total = a.B * (int64_t)a.S;   // S >= 1: covers the B trajectories too
    if (total <= 0) return hipSuccess;
    const unsigned blocks = (unsigned)((total + 255) / 256);
    if (f32) hipLaunchKernelGGL(time_alloc_init_kernel<float>, dim3(blocks), dim3(256), 0, st, a, total, vw, iters, done, pending, vw0);
    else hipLaunchKernelGGL(time_alloc_init_kernel<double>, dim3(blocks), dim3(256), 0, st, a, total, vw, iters, done, pending, vw0);
    return hipGetLastError();
}

hipError_t launch_time_alloc(const TimeAllocArgs &a, bool f32, hipStream_t st) {
    int64_t total = 0;
    if (a.seg_off) {
        // total segment count lives on the device for ragged device-memory callers
        hipError_t e = hipMemcpyAsync(&total, a.seg_off + a.B, sizeof(int64_t), hipMemcpyDeviceToHost, st);
        if (e != hipSuccess) return e;
        e = hipStreamSynchronize(st);
        if (e != hipSuccess) return e;
    } else {
        total = a.B * (int64_t)a.S;
    }
    if (total <= 0) return hipSuccess;
    const int threads = 256;
    const unsigned blocks = (unsigned)((total + threads - 1) / threads);
    if (f32) hipLaunchKernelGGL(time_alloc_kernel<float>, dim3(blocks), dim3(threads), 0, st, a, total);
    else hipLaunchKernelGGL(time_alloc_kernel<double>, dim3(blocks), dim3(threads), 0, st, a, total);
    return hipGetLastError();
}

}  // namespace csp
