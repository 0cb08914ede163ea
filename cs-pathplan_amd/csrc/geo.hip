// geo.hip -- batched WGS84 <-> ECEF <-> ENU (reference: uavPathPlanning.cpp:893-1108).
// One lane per point; 24 B in, 24 B out, a handful of fp64 transcendentals in between: the kernel
// is trig-bound on small batches and HBM-bound on large ones.  The reference point's ECEF image
// and rotation are computed once on the host side of the launch and passed by value.
#include "../../include/csp_geo.h"
#include "../../include/csp_minsnap.h"

#include <hip/hip_runtime.h>
#include "minsnap_hoststage.h"
#include <cmath>

namespace {

constexpr double kA = 6378137.0;            // WGS84_A  (uavPathPlanning.hpp:134)
constexpr double kE2 = 0.006694379990141;   // WGS84_E2 (uavPathPlanning.hpp:135)
constexpr double kPi = 3.14159265358979323846;

struct RefFrame {
    double ecef[3];
    double cl, sl, co, so;  // cos/sin of the reference latitude and longitude
};

__host__ __device__ inline double prime_vertical_radius(double lat) {
    const double s = sin(lat);
    return kA / sqrt(1.0 - kE2 * s * s);
}

__host__ __device__ inline void lla_to_ecef(double lon_deg, double lat_deg, double alt, double (&e)[3]) {
    const double lat = lat_deg * kPi / 180.0, lon = lon_deg * kPi / 180.0;
    const double N = prime_vertical_radius(lat);
    const double cl = cos(lat), sl = sin(lat);
    e[0] = (N + alt) * cl * cos(lon);
    e[1] = (N + alt) * cl * sin(lon);
    e[2] = (N * (1 - kE2) + alt) * sl;
}

__global__ void __launch_bounds__(256) wgs84_to_enu_kernel(const double *lla, RefFrame f, double *enu, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double t[3];
    lla_to_ecef(lla[3 * i], lla[3 * i + 1], lla[3 * i + 2], t);
    const double dx = t[0] - f.ecef[0], dy = t[1] - f.ecef[1], dz = t[2] - f.ecef[2];
    enu[3 * i + 0] = -f.so * dx + f.co * dy + 0.0 * dz;
    enu[3 * i + 1] = -f.sl * f.co * dx + -f.sl * f.so * dy + f.cl * dz;
    enu[3 * i + 2] = f.cl * f.co * dx + f.cl * f.so * dy + f.sl * dz;
}

__global__ void __launch_bounds__(256) enu_to_wgs84_kernel(const double *enu, RefFrame f, double *lla, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double qe = enu[3 * i], qn = enu[3 * i + 1], qu = enu[3 * i + 2];
    const double x = f.ecef[0] + (-f.so * qe + -f.sl * f.co * qn + f.cl * f.co * qu);
    const double y = f.ecef[1] + (f.co * qe + -f.sl * f.so * qn + f.cl * f.so * qu);
    const double z = f.ecef[2] + (0.0 * qe + f.cl * qn + f.sl * qu);
    const double p = sqrt(x * x + y * y);
    const double theta = atan2(z * kA, p * kA * (1 - kE2));
    const double st = sin(theta), ct = cos(theta);
    double lat = atan2(z + kE2 * kA * (1 - kE2) * (st * st * st) / (1 - kE2), p - kE2 * kA * (ct * ct * ct));
    for (int it = 0; it < 10; ++it) {  // the reference's fixed-point refinement, tolerance 1e-12 rad
        const double N = prime_vertical_radius(lat);
        const double alt = p / cos(lat) - N;
        const double nl = atan2(z, p * (1 - kE2 * N / (N + alt)));
        const bool done = fabs(nl - lat) < 1e-12;
        lat = nl;
        if (done) break;
    }
    const double N = prime_vertical_radius(lat);
    lla[3 * i + 0] = atan2(y, x) * 180.0 / kPi;
    lla[3 * i + 1] = lat * 180.0 / kPi;
    lla[3 * i + 2] = (p < 1e-12) ? fabs(z) - kA * sqrt(1 - kE2) : p / cos(lat) - N;
}

RefFrame make_frame(const double *ref) {
    RefFrame f;
    lla_to_ecef(ref[0], ref[1], ref[2], f.ecef);
    const double lat = ref[1] * kPi / 180.0, lon = ref[0] * kPi / 180.0;
    f.cl = std::cos(lat); f.sl = std::sin(lat); f.co = std::cos(lon); f.so = std::sin(lon);
    return f;
}

template <typename K>
int run(K kernel, const double *in, const double *ref, double *out, int64_t n, uint32_t mem_space, int32_t device_id, void *stream) {
    if (n < 0 || (n > 0 && (!in || !out)) || !ref) return CSP_ERR_INVALID_ARG;
    if (n == 0) return CSP_OK;
    if (csp_minsnap_device_count() < 1) return CSP_ERR_NO_DEVICE;
    if (device_id >= 0 && hipSetDevice(device_id) != hipSuccess) return CSP_ERR_HIP;
    hipStream_t st = (hipStream_t)stream;
    const RefFrame f = make_frame(ref);
    const unsigned blocks = (unsigned)((n + 255) / 256);
    if (mem_space == CSP_MEM_DEVICE) {
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, st, in, f, out, n);
        return hipGetLastError() == hipSuccess ? CSP_OK : CSP_ERR_HIP;
    }
    // host memory: through the device's cached staging arena (minsnap_hoststage.h), synchronous
    const size_t bytes = (size_t)n * 24;
    int cur = 0;
    (void)hipGetDevice(&cur);
    csp::HostCall hc(cur, st);
    const size_t o_in = hc.in(in, bytes), o_out = hc.out(out, bytes);
    if (hc.upload() != hipSuccess) return CSP_ERR_HIP;
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, st, hc.ptr<const double>(o_in), f, hc.ptr<double>(o_out), n);
    if (hipGetLastError() != hipSuccess || hc.download() != hipSuccess) return CSP_ERR_HIP;
    return CSP_OK;
}

}  // namespace

extern "C" int csp_geo_wgs84_to_enu_batch(const double *lla, const double *ref_host, double *enu, int64_t n,
                                          uint32_t mem_space, int32_t device_id, void *hip_stream) {
    return run(wgs84_to_enu_kernel, lla, ref_host, enu, n, mem_space, device_id, hip_stream);
}

extern "C" int csp_geo_enu_to_wgs84_batch(const double *enu, const double *ref_host, double *lla, int64_t n,
                                          uint32_t mem_space, int32_t device_id, void *hip_stream) {
    return run(enu_to_wgs84_kernel, enu, ref_host, lla, n, mem_space, device_id, hip_stream);
}
