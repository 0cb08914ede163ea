// Does RCCL accept the same device twice in ncclCommInitAll (two ranks on one GPU)?  If so, the RCCL transport of
// csp_minsnap_solve_batch_sharded can be exercised for real on a one-GPU box.  Build: hipcc tools/rccl_dup_probe.cpp -lrccl
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <cstdio>
#include <vector>
int main() {
    int devs[2] = {0, 0};
    ncclComm_t comms[2];
    ncclResult_t r = ncclCommInitAll(comms, 2, devs);
    std::printf("ncclCommInitAll({0,0}) -> %d (%s)\n", (int)r, ncclGetErrorString(r));
    if (r != ncclSuccess) return 1;
    hipStream_t s[2];
    double *a, *b;
    hipSetDevice(0);
    hipStreamCreateWithFlags(&s[0], hipStreamNonBlocking);
    hipStreamCreateWithFlags(&s[1], hipStreamNonBlocking);
    const size_t n = 1 << 20;
    hipMalloc(&a, n * 8); hipMalloc(&b, n * 8);
    std::vector<double> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = (double)i * 0.5;
    hipMemcpy(a, h.data(), n * 8, hipMemcpyHostToDevice);
    hipMemset(b, 0, n * 8);
    ncclGroupStart();
    r = ncclSend(a, n * 8, ncclChar, 1, comms[0], s[0]);
    std::printf("send -> %d\n", (int)r);
    r = ncclRecv(b, n * 8, ncclChar, 0, comms[1], s[1]);
    std::printf("recv -> %d\n", (int)r);
    r = ncclGroupEnd();
    std::printf("groupEnd -> %d (%s)\n", (int)r, ncclGetErrorString(r));
    hipStreamSynchronize(s[0]); hipStreamSynchronize(s[1]);
    std::vector<double> g(n);
    hipMemcpy(g.data(), b, n * 8, hipMemcpyDeviceToHost);
    size_t bad = 0;
    for (size_t i = 0; i < n; ++i) bad += g[i] != h[i];
    std::printf("mismatches: %zu\n", bad);
    return bad ? 2 : 0;
}
