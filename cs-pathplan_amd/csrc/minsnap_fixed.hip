// minsnap_fixed.hip -- dispatcher of the register-resident fixed-size kernels.  The kernels live
// in minsnap_fixed_impl.h and are instantiated per derivative order in minsnap_fixed_o<O>.hip
// (separate translation units: they compile in parallel).
//
// Buckets served: fp64, uniform S (either parity), and
//   order 2, 3 : 2 <= S <= 16   (min-acceleration = the reference's shipped yaml, min-jerk = its default)
//   order 4    : 2 <= S <= 16, both coefficient layouts (the headline minimum-snap bucket)
//   order 5    : 2 <= S <= 8    (4x4 blocks: 28 doubles per waypoint stay in registers up to 3 waypoints per half)
// With the path-deviation penalty (path_weight > 0): orders 2..4, 2 <= S <= 16, default layout
// (minsnap_fixed_path_impl.h: pre-solve, t* pick and penalised solve in one launch).
// Everything else goes to the generic kernel.
#include "minsnap_launch.h"

namespace csp {

hipError_t launch_fixed_o2(const GenericArgs &a, int cus, hipStream_t st);
hipError_t launch_fixed_o3(const GenericArgs &a, int cus, hipStream_t st);
hipError_t launch_fixed_o4a(const GenericArgs &a, int cus, hipStream_t st);   // S = 2..9
hipError_t launch_fixed_o4b(const GenericArgs &a, int cus, hipStream_t st);   // S = 10..16
hipError_t launch_fixed_o5(const GenericArgs &a, int cus, hipStream_t st);
hipError_t launch_fixedpath_o2(const GenericArgs &a, hipStream_t st);
hipError_t launch_fixedpath_o3(const GenericArgs &a, hipStream_t st);
hipError_t launch_fixedpath_o4a(const GenericArgs &a, hipStream_t st);  // S = 2..9
hipError_t launch_fixedpath_o4b(const GenericArgs &a, hipStream_t st);  // S = 10..16

bool fixed_supported(int order, int S, bool f32, double path_weight, bool ragged, bool seg_major) {
    if (f32 || ragged || S < 2) return false;
    if (path_weight != 0.0) return path_weight > 0.0 && order >= 2 && order <= 4 && S <= 16 && !seg_major;
    switch (order) {
        case 2: case 3: return S <= 16 && !seg_major;
        case 4: return S <= 16;
        case 5: return S <= 8 && !seg_major;
    }
    return false;
}

const char *fixed_kernel_name(int order, int S, bool path) {
    static thread_local char name[32];
    std::snprintf(name, sizeof name, "fixed%s_o%d_s%d_f64", path ? "path" : "", order, S);
    return name;
}

hipError_t launch_fixed(const GenericArgs &a, hipStream_t st) {
    if (a.B == 0 && !a.multi) return hipSuccess;
    if (a.path_weight > 0.0) {
        // the kernel writes status and max_dev itself, and only for trajectories not marked in `skip`
        switch (a.order) {
            case 2: return launch_fixedpath_o2(a, st);
            case 3: return launch_fixedpath_o3(a, st);
            case 4: return a.S <= 9 ? launch_fixedpath_o4a(a, st) : launch_fixedpath_o4b(a, st);
        }
        return hipErrorInvalidValue;
    }
    hipError_t e;
    if (a.status && !a.multi && (e = hipMemsetAsync(a.status, 0, sizeof(int32_t) * (size_t)a.B, st)) != hipSuccess) return e;
    // without the path penalty the reference's deviation metric is evaluated at t* = 0, where the
    // polynomial equals its waypoint exactly (minimum_snap.cpp:342, :596-617)
    if (a.max_dev && !a.multi && (e = hipMemsetAsync(a.max_dev, 0, sizeof(double) * (size_t)a.B, st)) != hipSuccess) return e;
    // persistent grid: two workgroups per CU (register- and LDS-limited residency of these kernels)
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    switch (a.order) {
        case 2: return launch_fixed_o2(a, cus, st);
        case 3: return launch_fixed_o3(a, cus, st);
        case 4: return a.S <= 9 ? launch_fixed_o4a(a, cus, st) : launch_fixed_o4b(a, cus, st);
        case 5: return launch_fixed_o5(a, cus, st);
    }
    return hipErrorInvalidValue;
}

}  // namespace csp
