// minsnap_chunked.hip -- workspace-free kernel for ragged and long trajectories (any S <= 256,
// orders 2..5, fp64 or fp32 storage with fp64 arithmetic, zero-velocity penalty, no path penalty).
// The device code is minsnap_chunked_impl.h (shared with the mixed-order entry, minsnap_mixed.hip); this file holds the
// one-order kernel and its launcher.
#include "minsnap_chunked_impl.h"

namespace csp {
namespace chunked {

// STATUS = the caller passed a `status` array: only then are the coefficients tested for NaN/Inf (on the values
// as stored, so that an fp32 overflow is caught: two conversions and an fma per coefficient otherwise spent for nothing).
template <int O, typename IO, bool STATUS>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(O <= 4 ? 2 : 1)))
minsnap_chunked_kernel(GenericArgs a, int lpt_log2) {
    using IL = IfaceLds<O>;
    __shared__ double lds[IL::ENTRIES * 64];
    __shared__ double xch[3 * (O - 1) * 64];   // solved interface derivatives, handed to the right-hand neighbour
    const int lane = threadIdx.x;
    const int64_t b = ((int64_t)blockIdx.x * 64 + lane) >> lpt_log2;  // my trajectory
    const bool traj_ok = b < a.B;
    const int64_t bb = traj_ok ? b : a.B - 1;
    int64_t seg0;
    int S;
    if (a.seg_off) { seg0 = a.seg_off[bb]; S = (int)(a.seg_off[bb + 1] - seg0); }
    else { seg0 = bb * (int64_t)a.S; S = a.S; }
    chunked_body<O, IO, STATUS, false>(a, lds, xch, lane, 1 << lpt_log2, lane & ((1 << lpt_log2) - 1), traj_ok, bb, seg0, S, seg0 * (int64_t)(6 * O));
}

template <int O> hipError_t launch_o(const GenericArgs &a, bool f32, int lpt_log2, hipStream_t st) {
    const int64_t lanes = a.B << lpt_log2;
    const dim3 grid((unsigned)((lanes + 63) / 64)), block(64);
    if (a.status) {
        if (f32) hipLaunchKernelGGL((minsnap_chunked_kernel<O, float, true>), grid, block, 0, st, a, lpt_log2);
        else hipLaunchKernelGGL((minsnap_chunked_kernel<O, double, true>), grid, block, 0, st, a, lpt_log2);
    } else {
        if (f32) hipLaunchKernelGGL((minsnap_chunked_kernel<O, float, false>), grid, block, 0, st, a, lpt_log2);
        else hipLaunchKernelGGL((minsnap_chunked_kernel<O, double, false>), grid, block, 0, st, a, lpt_log2);
    }
    return hipGetLastError();
}

}  // namespace chunked

int chunked_lanes_log2(int Smax) {
    int l = 0;
    while ((chunked::CMAX << l) < Smax) ++l;
    return l;
}

bool chunked_supported(int order, int Smax, bool f32_arith, double path_weight, bool seg_major) {
    return order >= 2 && order <= 5 && Smax >= 1 && Smax <= chunked::CMAX * 64 && !f32_arith && path_weight == 0.0 && !seg_major;
}

hipError_t launch_chunked(const GenericArgs &a, bool f32, int Smax, hipStream_t st) {
    if (a.B == 0) return hipSuccess;
    hipError_t e;
    if (a.status && (e = hipMemsetAsync(a.status, 0, sizeof(int32_t) * (size_t)a.B, st)) != hipSuccess) return e;
    // no path penalty: the deviation metric is evaluated at t* = 0 where it vanishes (minimum_snap.cpp:342)
    if (a.max_dev && (e = hipMemsetAsync(a.max_dev, 0, sizeof(double) * (size_t)a.B, st)) != hipSuccess) return e;
    const int l = chunked_lanes_log2(Smax);
    switch (a.order) {
        case 2: return chunked::launch_o<2>(a, f32, l, st);
        case 3: return chunked::launch_o<3>(a, f32, l, st);
        case 4: return chunked::launch_o<4>(a, f32, l, st);
        case 5: return chunked::launch_o<5>(a, f32, l, st);
    }
    return hipErrorInvalidValue;
}

}  // namespace csp
