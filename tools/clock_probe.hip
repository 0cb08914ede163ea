// clock_probe.hip -- what one lone wave gets: a chain of N dependent fp64 FMAs (and, separately, N dependent
// v_readlane -> VALU round trips) timed with HIP events after an idle gap, the situation of the one-flight kernels.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/clock_probe tools/clock_probe.hip && /tmp/clock_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>

__global__ void fma_chain(double *out, int n, double a, double b) {
    double x = out[blockIdx.x * 64 + threadIdx.x];
#pragma unroll 50
    for (int i = 0; i < n; ++i) x = __builtin_fma(x, a, b);
    out[blockIdx.x * 64 + threadIdx.x] = x;
}
__global__ void lane_chain(double *out, int n) {
    double x = out[threadIdx.x];
    int l = threadIdx.x & 63;
#pragma unroll 50
    for (int i = 0; i < n; ++i) {
        const unsigned lo = __builtin_amdgcn_readlane((unsigned)__double_as_longlong(x), (l + i) & 63);
        x = x * 0.5 + (double)lo * 1e-30;
        l = __builtin_amdgcn_readfirstlane(l + 1);
    }
    out[threadIdx.x] = x;
}

int main() {
    double *d;
    hipMalloc(&d, 64 * 8);
    hipMemset(d, 0, 64 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    // the same chain with the whole chip busy (one wave per SIMD, every wave its own chain): per-step time is a latency, so
    // it can only drop if the CLOCK is higher under load
    for (int blocks : {1, 256, 1024}) {
        double *big;
        hipMalloc(&big, (size_t)blocks * 64 * 8);
        hipMemset(big, 0, (size_t)blocks * 64 * 8);
        const int n = 20000;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0);
            fma_chain<<<blocks, 64>>>(big, n, 0.999, 1e-3);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (rep == 3) std::printf("{\"blocks\": %d, \"chain\": \"fma_f64\", \"ns_per_step\": %.2f}\n", blocks, ms * 1e6 / n);
        }
        hipFree(big);
    }
    for (int idle_ms : {0, 20}) {
        for (int which = 0; which < 2; ++which) {
            const int n = 20000;
            for (int rep = 0; rep < 3; ++rep) {
                std::this_thread::sleep_for(std::chrono::milliseconds(idle_ms));
                hipEventRecord(e0);
                if (which == 0) fma_chain<<<1, 64>>>(d, n, 0.999, 1e-3);
                else lane_chain<<<1, 64>>>(d, n);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (rep == 2)
                    std::printf("{\"idle_ms\": %d, \"chain\": \"%s\", \"n\": %d, \"us\": %.1f, \"ns_per_step\": %.2f}\n", idle_ms,
                                which ? "readlane+fma" : "fma_f64", n, ms * 1e3, ms * 1e6 / n);
            }
        }
    }
    return 0;
}
