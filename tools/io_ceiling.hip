// io_ceiling.hip -- what does the MEMORY SYSTEM allow for the headline kernel's traffic, with the arithmetic taken out?
// Same shape as minsnap_fixed_persistent_kernel<4,16>: 2 x CU persistent workgroups of 128 threads walk 64-trajectory
// slices (and, for comparison, 4 / 8 / 16 workgroups per CU: more waves = more bytes in flight than the solver can have); per slice a workgroup reads the slice's 34 816 B of inputs (16 B per lane, coalesced) and writes its 196 608 B
// of coefficients -- either in the kernel's own store shape (one wave instruction = 4 trajectory rows x 256 contiguous
// bytes, rows 3072 B apart: "rows") or as fully contiguous 1 KB per wave instruction ("linear").  The written values
// depend on what was read, so nothing is optimised away.
//   hipcc --offload-arch=gfx950 -O3 tools/io_ceiling.hip -o /tmp/io_ceiling && /tmp/io_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef double v2d_t __attribute__((ext_vector_type(2)));

// NT: the stores carry the non-temporal bit (what the solver does for launches beyond the Infinity Cache)
template <bool ROWS, bool NT>
__global__ void __launch_bounds__(128) io_kernel(const double2 *in, double2 *out, int n_slices) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int s = blockIdx.x; s < n_slices; s += gridDim.x) {
        const double2 *src = in + (size_t)s * (34816 / 16);
        double2 acc = make_double2(0.0, 0.0);
        for (int i = tid; i < 34816 / 16; i += 128) { const double2 v = src[i]; acc.x += v.x; acc.y += v.y; }
        char *dst = (char *)out + (size_t)s * 196608;
        // each wave writes half of the slice: 96 wave-instructions of 1 KB
        for (int k = 0; k < 96; ++k) {
            size_t off;
            if (ROWS) {
                // instruction k of wave w: segment pair q = k / 12 (of 8 per half... 8 pairs x 12 instructions), rows 4*(k%12.. ) -- 16 row groups x 6 pieces
                const int pair = (k / 16) + wave * 4 + (k / 16 >= 4 ? 0 : 0);   // 6 pairs per wave half: k/16 in 0..5
                const int rowgrp = k % 16;                                        // 16 groups of 4 trajectories
                off = (size_t)(rowgrp * 4 + (lane >> 4)) * 3072 + (size_t)((pair % 8) * 384) + (size_t)(lane & 15) * 16;
            } else {
                off = (size_t)(wave * 96 + k) * 1024 + (size_t)lane * 16;
            }
            acc.x += 1.0;
            if (NT) { v2d_t x = {acc.x, acc.y}; __builtin_nontemporal_store(x, reinterpret_cast<v2d_t *>(dst + off)); }
            else *reinterpret_cast<double2 *>(dst + off) = acc;
        }
    }
}

int main() {
    const int B = 524288, n_slices = B / 64;
    double2 *in, *out;
    hipMalloc(&in, (size_t)n_slices * 34816);
    hipMalloc(&out, (size_t)n_slices * 196608);
    hipMemset(in, 0, (size_t)n_slices * 34816);
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int nt = 0; nt < 2; ++nt)
    for (int per_cu = 2; per_cu <= 8; per_cu *= 2)
    for (int mode = 0; mode < 2; ++mode) {
        const int grid = per_cu * p.multiProcessorCount;
        auto launch = [&] {
            if (mode) { if (nt) hipLaunchKernelGGL((io_kernel<true, true>), dim3(grid), dim3(128), 0, 0, in, out, n_slices);
                        else hipLaunchKernelGGL((io_kernel<true, false>), dim3(grid), dim3(128), 0, 0, in, out, n_slices); }
            else { if (nt) hipLaunchKernelGGL((io_kernel<false, true>), dim3(grid), dim3(128), 0, 0, in, out, n_slices);
                   else hipLaunchKernelGGL((io_kernel<false, false>), dim3(grid), dim3(128), 0, 0, in, out, n_slices); }
        };
        for (int w = 0; w < 3; ++w) launch();
        hipEventRecord(e0, 0);
        const int reps = 20;
        for (int r = 0; r < reps; ++r) launch();
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        ms /= reps;
        const double bytes = (double)n_slices * (34816.0 + 196608.0);
        std::printf("{\"nt_stores\": %d, \"workgroups_per_cu\": %d, \"store_shape\": \"%s\", \"B\": %d, \"us\": %.1f, \"TBps\": %.3f, \"frac_of_8TBps\": %.3f}\n", nt, per_cu, mode ? "rows (4 x 256 B per wave instruction)" : "linear (1 KB per wave instruction)",
                    B, ms * 1e3, bytes / (ms * 1e-3) / 1e12, bytes / (ms * 1e-3) / 8e12);
    }
    return 0;
}
