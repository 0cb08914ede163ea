// minsnap_mixed.hip -- mixed-ORDER ragged batches in one call (BASELINE config 5: per-trajectory segment count and
// derivative order; csp_minsnap_solve_mixed, include/csp_minsnap.h).  Everything happens on the device and in the
// caller's own order -- no gather, no un-permute (blocks 16-byte aligned: block_elems):
//
//   1. bucketing (three small kernels): every trajectory gets the key (order, length class) -- the length class is the
//      number of lanes the workspace-free kernel gives a trajectory (minsnap_chunked_impl.h: 4 segments per lane, rounded
//      up to a power of two) --; a histogram, an exclusive scan of the coefficient sizes 6 * order * S in CALLER order
//      (= where each trajectory's block starts in `coeffs`) and a scatter of the trajectory indices into bucket order
//      (`perm`, longest class first inside an order);
//   2. one PERSISTENT launch per derivative order (the register budgets differ: 190 / 256 / 392 VGPRs at orders 3 / 4 /
//      5, so one kernel for all would run everything at one wave per SIMD): wave w takes the work units w, w + grid, ...
//      of its order -- a unit = 64 lanes of one length class --, reading the inputs and writing the coefficients of
//      trajectory perm[k] in place.  The launches of the orders run concurrently on forked streams.
//
// The reference has no batched entry at all (one flight per call, uavPathPlanning.cpp:4423, :4461); per trajectory the
// arithmetic is that of csp_minsnap_solve_batch on the same order and length class (bit-equal: tests/test_gpu_round3.py).
#include "minsnap_chunked_impl.h"
#include "minsnap_mixed.h"
#include "minsnap_twist_launch.h"

#include <cstdlib>

namespace csp {
namespace mixed {

static __constant__ TwistCostOrder g_cost_order = make_twist_cost_order();

constexpr int NCLS = MIXED_NCLS;
constexpr int NORD = 4;          // orders 2..5
constexpr int NGRP = MIXED_NGRP; // (family, order): family 0 = the lane-pair sweep (S <= 64), family 1 = the chunked kernel (65 .. 256)
constexpr int NKEY = NGRP * NCLS;
constexpr int ITEMS = 4;         // trajectories per thread of the bucketing kernels
constexpr int BT = 256;          // threads per bucketing block
constexpr int PT = NKEY;         // threads of the planning block: one per (group, class) key
static_assert(PT == 512, "the planning block's scans are written for eight waves");

// lanes a trajectory of S segments gets: one per chunk of <= 4 segments.  (The one-order kernel rounds this up to a power
// of two and gives a whole call the lanes of its longest trajectory; here the classes are exact, so a 33-segment trajectory
// takes 9 lanes, not 16, seven of them share a wave, and the interface solve runs 7 steps, not 14.)
__device__ __forceinline__ int lanes_of(int S) { return (S + chunked::CMAX - 1) / chunked::CMAX; }

// Elements of a trajectory's coefficient block in the caller-order layout: 6 * order * S, rounded up to a whole number of
// 16-byte pieces (fp32 storage: a multiple of 4 elements -- odd orders with an odd segment count carry 2 floats of padding),
// so that every block starts 16-byte aligned and records leave in 16-byte stores (8-byte stores to 8-byte aligned blocks
// doubled the write requests and cost the fp32 mixed batch a fifth of its time).
__device__ __forceinline__ long long block_elems(int o, int64_t S, int pad_to) {
    if (o < 1 || S <= 0) return 0;
    const long long n = (long long)S * 6 * o;
    return (n + pad_to - 1) / pad_to * pad_to;
}

// key = group * NCLS + class, or -1 for a trajectory this path does not serve (order outside 2..5, S outside 1..smax).
// twist_mask bit (order - 2): trajectories of up to 64 segments of that order go to the lane-pair sweep.
__device__ __forceinline__ int key_of(int order, int64_t S, int smax, int twist_mask) {
    if (order < 2 || order > 5 || S < 1 || S > chunked::CMAX * 64 || S > smax) return -1;
    if (S <= twist::SMAX && ((twist_mask >> (order - 2)) & 1)) return (order - 2) * NCLS + (twist::SMAX - (int)S);
    return (NORD + order - 2) * NCLS + (64 - lanes_of((int)S));
}

// No global atomics anywhere in the bucketing: same-address device-scope atomics cost ~60 ns EACH on this part (measured:
// a first version whose persistent waves pulled work units from one counter spent 1.2 ms on 20 k atomicAdds, whatever
// the work), so blocks publish their histograms and the planning block turns them into per-block offsets.
__global__ void __launch_bounds__(BT) count_kernel(const int32_t *orders, const int64_t *seg_off, int64_t B, int32_t *hist_blk, int64_t *block_sum, int pad_to,
                                                   int smax, int twist_mask) {
    __shared__ int hist[NKEY];
    __shared__ long long wsum[BT / 64];
    const int tid = threadIdx.x;
    for (int q = tid; q < NKEY; q += BT) hist[q] = 0;
    __syncthreads();
    long long csz = 0;
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        const int64_t i = ((int64_t)blockIdx.x * ITEMS + it) * BT + tid;
        if (i < B) {
            const int o = orders[i];
            const int64_t S = seg_off[i + 1] - seg_off[i];
            const int k = key_of(o, S, smax, twist_mask);
            if (k >= 0) atomicAdd(&hist[k], 1);
            // the coefficient block exists in the caller's layout whether or not the trajectory is served
            csz += block_elems(o, S, pad_to);
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) csz += __shfl_xor(csz, d);
    if ((tid & 63) == 0) wsum[tid >> 6] = csz;
    __syncthreads();
    for (int q = tid; q < NKEY; q += BT) hist_blk[(int64_t)blockIdx.x * NKEY + q] = hist[q];
    if (tid == 0) {
        long long s = 0;
        for (int w = 0; w < BT / 64; ++w) s += wsum[w];
        block_sum[blockIdx.x] = s;
    }
}

// one block: bucket starts / work units per order, exclusive scan of the per-block coefficient sizes
__global__ void __launch_bounds__(PT) plan_kernel(MixedTable *tab, int32_t *hist_blk, int64_t *block_sum, int64_t nblk, int64_t *coef_off, int64_t B) {
    __shared__ long long carry;
    __shared__ long long wtot[PT / 64];
    __shared__ int total[NKEY];
    const int tid = threadIdx.x;
    // per key: exclusive scan over the blocks of that key's counts (in place: hist_blk becomes the block's offset inside its
    // bucket); loads batched 64 at a time
    {
        int run = 0;
        for (int64_t b0 = 0; b0 < nblk; b0 += 64) {   // 64 loads in flight per round trip (B = 65536: one round trip)
            int v[64];
#pragma unroll
            for (int q = 0; q < 64; ++q) v[q] = b0 + q < nblk ? hist_blk[(b0 + q) * NKEY + tid] : 0;
#pragma unroll
            for (int q = 0; q < 64; ++q) {
                if (b0 + q < nblk) hist_blk[(b0 + q) * NKEY + tid] = run;
                run += v[q];
            }
        }
        total[tid] = run;
        tab->count[tid] = run;
    }
    if (tid == 0) carry = 0;
    // bucket starts (exclusive scan of the 512 totals, groups back to back) and work units per group (one group = the 64
    // classes = one wave: a wave-level scan).  In parallel: one thread walking the entries with a global store each took
    // 36 us -- an eighth of the C5 call.
    {
        const int o = tid >> 6, k = tid & 63;          // PT = NGRP * NCLS: thread = key, wave = group
        const int n = total[tid];
        const int per_wave = o < NORD ? 64 : 64 / (64 - k);   // trajectories per work unit
        const int units = (n + per_wave - 1) / per_wave;
        int xn = n, xu = units;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int yn = __shfl_up(xn, d), yu = __shfl_up(xu, d);
            if (k >= d) { xn += yn; xu += yu; }
        }
        __shared__ int wave_n[NGRP];
        if (k == 63) wave_n[o] = xn;
        __syncthreads();
        int before = 0;                                 // trajectories of the orders before mine
        for (int w = 0; w < o; ++w) before += wave_n[w];
        tab->bucket_start[o][k] = before + xn - n;
        tab->unit_start[o][k] = xu - units;
        if (k == 63) {
            tab->bucket_start[o][NCLS] = before + xn;
            tab->unit_start[o][NCLS] = xu;
            if (o == NGRP - 1) tab->served = before + xn;
        }
    }
    // the lane-pair sweep's units in cost order (threads 0..255 = positions of that order: four waves)
    {
        __shared__ int wave_u[4];
        const int key = tid < 4 * NCLS ? g_cost_order.key_at[tid] : 0;
        const int units = tid < 4 * NCLS ? (total[key] + 63) / 64 : 0;
        int x = units;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int y = __shfl_up(x, d);
            if ((tid & 63) >= d) x += y;
        }
        if (tid < 4 * NCLS && (tid & 63) == 63) wave_u[tid >> 6] = x;
        __syncthreads();
        if (tid < 4 * NCLS) {
            int before = 0;
            for (int w = 0; w < (tid >> 6); ++w) before += wave_u[w];
            tab->tw_ustart[tid] = before + x - units;
            if (tid == 4 * NCLS - 1) tab->tw_ustart[4 * NCLS] = before + x;
        }
        if (tid == 0) tab->next_unit = 0;
    }
    __syncthreads();
    // exclusive scan of block_sum, PT entries per round
    for (int64_t base = 0; base < nblk; base += PT) {
        const int64_t i = base + tid;
        long long v = i < nblk ? block_sum[i] : 0, x = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const long long y = __shfl_up(x, d);
            if ((tid & 63) >= d) x += y;
        }
        if ((tid & 63) == 63) wtot[tid >> 6] = x;
        __syncthreads();
        long long pre = carry;
        for (int w = 0; w < (tid >> 6); ++w) pre += wtot[w];
        if (i < nblk) block_sum[i] = pre + x - v;
        __syncthreads();
        if (tid == PT - 1) carry = pre + x;
        __syncthreads();
    }
    if (tid == 0) coef_off[B] = carry;
}

__global__ void __launch_bounds__(BT) scatter_kernel(const int32_t *orders, const int64_t *seg_off, int64_t B, const MixedTable *tab,
                                                     const int32_t *blk_base, const int64_t *block_pre, int64_t *coef_off, int32_t *perm,
                                                     int32_t *status, int pad_to, int smax, int twist_mask) {
    __shared__ int hist[NKEY], base[NKEY];
    __shared__ long long wtot[BT / 64];
    const int tid = threadIdx.x;
    for (int q = tid; q < NKEY; q += BT) hist[q] = 0;
    __syncthreads();
    int key[ITEMS], rank[ITEMS];
    long long csz[ITEMS], mine = 0;
    // thread t owns the ITEMS consecutive trajectories (blk * BT + t) * ITEMS .. : the scan of the coefficient sizes is in
    // caller order
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        const int64_t i = ((int64_t)blockIdx.x * BT + tid) * ITEMS + it;
        key[it] = -1;
        csz[it] = 0;
        rank[it] = 0;
        if (i < B) {
            const int o = orders[i];
            const int64_t S = seg_off[i + 1] - seg_off[i];
            key[it] = key_of(o, S, smax, twist_mask);
            csz[it] = block_elems(o, S, pad_to);
            if (key[it] >= 0) rank[it] = atomicAdd(&hist[key[it]], 1);
            if (status) status[i] = key[it] >= 0 ? 0 : CSP_TRAJ_SKIPPED_BIT;   // the solve kernels OR their bits in
        }
        mine += csz[it];
    }
    long long x = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const long long y = __shfl_up(x, d);
        if ((tid & 63) >= d) x += y;
    }
    if ((tid & 63) == 63) wtot[tid >> 6] = x;
    __syncthreads();
    for (int q = tid; q < NKEY; q += BT) base[q] = blk_base[(int64_t)blockIdx.x * NKEY + q];
    long long pre = block_pre[blockIdx.x] + x - mine;
    for (int w = 0; w < (tid >> 6); ++w) pre += wtot[w];
    __syncthreads();
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        const int64_t i = ((int64_t)blockIdx.x * BT + tid) * ITEMS + it;
        if (i < B) {
            coef_off[i] = pre;
            pre += csz[it];
            if (key[it] >= 0) {
                const int o = key[it] / NCLS, k = key[it] - o * NCLS;
                perm[tab->bucket_start[o][k] + base[key[it]] + rank[it]] = (int32_t)i;
            }
        }
    }
}

// Persistent solve of ONE order's buckets: wave w of the grid takes the work units w, w + gridDim.x, ... of this order (a
// unit = 64 lanes of one length class; the list runs longest class first, so the strided deal is balanced) -- a fixed trip
// count every wave reaches, no work counter (see count_kernel for what one costs).
template <int O, typename IO, bool STATUS>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(O <= 4 ? 2 : 1)))
minsnap_chunked_mixed_kernel(GenericArgs a, const int32_t *perm, const int64_t *coef_off, const MixedTable *tab) {
    using IL = iface::IfaceLds<O>;
    __shared__ double lds[IL::ENTRIES * 64];
    __shared__ double xch[3 * (O - 1) * 64];
    __shared__ int s_ustart[NCLS + 1], s_bstart[NCLS + 1];
    const int lane = threadIdx.x;
    constexpr int oi = NORD + O - 2;   // group: chunked family, this order
    s_ustart[lane] = tab->unit_start[oi][lane];
    s_bstart[lane] = tab->bucket_start[oi][lane];
    if (lane == 0) { s_ustart[NCLS] = tab->unit_start[oi][NCLS]; s_bstart[NCLS] = tab->bucket_start[oi][NCLS]; }
    __syncthreads();
    const int total = s_ustart[NCLS];
    for (int u = blockIdx.x; u < total; u += gridDim.x) {
        // the class of unit u: the last k with unit_start[k] <= u (binary search, wave-uniform LDS broadcasts)
        int k = 0;
#pragma unroll
        for (int step = NCLS / 2; step >= 1; step >>= 1)
            if (s_ustart[k + step] <= u) k += step;
        const int lpt = 64 - k;                      // lanes per trajectory
        const int tpw = 64 / lpt;                    // trajectories per wave
        const int lo = s_bstart[k], hi = s_bstart[k + 1];
        // lane -> (trajectory of the wave, chunk): exact for lane < 64 (multiply-shift reciprocal)
        const int t_in = (lane * ((65536 + lpt - 1) / lpt)) >> 16;
        const int j = lane - t_in * lpt;
        const int idx = lo + (u - s_ustart[k]) * tpw + t_in;
        const bool traj_ok = t_in < tpw && idx < hi;
        const int64_t bb = perm[traj_ok ? idx : hi - 1];
        const int64_t seg0 = a.seg_off[bb];
        const int S = (int)(a.seg_off[bb + 1] - seg0);
        chunked::chunked_body<O, IO, STATUS, false>(a, lds, xch, lane, lpt, j, traj_ok, bb, seg0, S, coef_off[bb]);
        fixedk::lds_barrier();   // the LDS images are reused by the next unit (LDS-only: the unit's stores stay in flight)
    }
}

template <int O> hipError_t launch_order(const GenericArgs &a, bool f32, const int32_t *perm, const int64_t *coef_off, const MixedTable *tab,
                                         int waves, hipStream_t st) {
    const dim3 grid((unsigned)waves), block(64);
    if (a.status) {
        if (f32) hipLaunchKernelGGL((minsnap_chunked_mixed_kernel<O, float, true>), grid, block, 0, st, a, perm, coef_off, tab);
        else hipLaunchKernelGGL((minsnap_chunked_mixed_kernel<O, double, true>), grid, block, 0, st, a, perm, coef_off, tab);
    } else {
        if (f32) hipLaunchKernelGGL((minsnap_chunked_mixed_kernel<O, float, false>), grid, block, 0, st, a, perm, coef_off, tab);
        else hipLaunchKernelGGL((minsnap_chunked_mixed_kernel<O, double, false>), grid, block, 0, st, a, perm, coef_off, tab);
    }
    return hipGetLastError();
}

}  // namespace mixed

// checkpoint slots of the lane-pair sweep: one per persistent workgroup, sized for the order that needs most at the
// caller's max_segments (a role of ceil(smax / 2) segments keeps one checkpoint per block boundary)
namespace {
constexpr int TWIST_MAX_WG = 512;
template <int O> size_t twist_role_doubles(int smax) {
    using G = twist::Geo<O>;
    const int hs = ((smax < twist::SMAX ? smax : twist::SMAX) + 1) / 2;
    const int nck = (hs + G::K - 1) / G::K - 1;
    return (size_t)(nck > 0 ? nck : 0) * G::CKD * 64;
}
size_t twist_slot_doubles_per_role(int smax) {
    size_t m = twist_role_doubles<2>(smax);
    if (twist_role_doubles<3>(smax) > m) m = twist_role_doubles<3>(smax);
    if (twist_role_doubles<4>(smax) > m) m = twist_role_doubles<4>(smax);
    if (twist_role_doubles<5>(smax) > m) m = twist_role_doubles<5>(smax);
    return m + 64;   // never empty
}
// a work unit holds at least one trajectory; every (order, S) class ends in at most one partial unit
int twist_slots(int64_t B) {
    int64_t n = B / 64 + mixed::NORD * mixed::NCLS;
    if (n > B) n = B;
    if (n > TWIST_MAX_WG) n = TWIST_MAX_WG;
    return (int)n;
}
}  // namespace

size_t mixed_workspace_bytes(int64_t B, int smax) {
    const int64_t nblk = (B + mixed::BT * mixed::ITEMS - 1) / (mixed::BT * mixed::ITEMS);
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    return up(sizeof(MixedTable)) + up((size_t)(nblk + 1) * 8) + up((size_t)nblk * mixed::NKEY * 4) + up((size_t)(B + 1) * 8) + up((size_t)B * 4) +
           up((size_t)twist_slots(B) * twist_slot_doubles_per_role(smax) * 2 * sizeof(double));
}

// Forked streams of one device for the per-order launches (created once per device and thread, never destroyed: the
// library has no teardown hook; four streams and eight events per device).
namespace {
struct Fork {
    hipStream_t s[3] = {nullptr, nullptr, nullptr};
    hipEvent_t begin = nullptr, done[3] = {nullptr, nullptr, nullptr};
    bool ok = false;
};
Fork *fork_for_device() {
    static thread_local Fork forks[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    Fork &f = forks[dev];
    if (!f.ok) {
        for (int i = 0; i < 3; ++i) {
            if (hipStreamCreateWithFlags(&f.s[i], hipStreamNonBlocking) != hipSuccess) return nullptr;
            if (hipEventCreateWithFlags(&f.done[i], hipEventDisableTiming) != hipSuccess) return nullptr;
        }
        if (hipEventCreateWithFlags(&f.begin, hipEventDisableTiming) != hipSuccess) return nullptr;
        f.ok = true;
    }
    return &f;
}
}  // namespace

hipError_t launch_mixed(const GenericArgs &a, bool f32, const int32_t *orders, void *workspace, int64_t *coef_off_out, hipStream_t st) {
    using namespace mixed;
    if (a.B == 0) return hipSuccess;
    const int64_t B = a.B;
    const int64_t nblk = (B + BT * ITEMS - 1) / (BT * ITEMS);
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    char *w = (char *)workspace;
    MixedTable *tab = (MixedTable *)w;                 w += up(sizeof(MixedTable));
    int64_t *block_sum = (int64_t *)w;                 w += up((size_t)(nblk + 1) * 8);
    int32_t *hist_blk = (int32_t *)w;                  w += up((size_t)nblk * NKEY * 4);
    int64_t *coef_ws = (int64_t *)w;                   w += up((size_t)(B + 1) * 8);
    int32_t *perm = (int32_t *)w;                      w += up((size_t)B * 4);
    double *ckws = (double *)w;
    int64_t *coef_off = coef_off_out ? coef_off_out : coef_ws;
    hipError_t e;
    // no memsets: the three kernels write every word they or the solve read (status included)
    const int pad_to = f32 ? 4 : 2;
    const int smax = a.S > 0 ? a.S : chunked::CMAX * 64;   // the caller's max_segments: longer trajectories are skipped
    // bit (order - 2): that order's trajectories of up to 64 segments go to the lane-pair sweep (minsnap_twist_impl.h), the
    // longer ones to the chunked kernels.  Default: all four orders; CSP_MIXED_TWIST=0 puts everything on the chunked kernels
    // (the round-3 baseline, tools/twist_probe.py measures one against the other)
    static const int twist_mask = [] { const char *e = std::getenv("CSP_MIXED_TWIST"); return e ? (int)std::strtol(e, nullptr, 0) & 15 : 15; }();
    hipLaunchKernelGGL(count_kernel, dim3((unsigned)nblk), dim3(BT), 0, st, orders, a.seg_off, B, hist_blk, block_sum, pad_to, smax, twist_mask);
    hipLaunchKernelGGL(plan_kernel, dim3(1), dim3(PT), 0, st, tab, hist_blk, block_sum, nblk, coef_off, B);
    hipLaunchKernelGGL(scatter_kernel, dim3((unsigned)nblk), dim3(BT), 0, st, orders, a.seg_off, B, tab, hist_blk, block_sum, coef_off, perm, a.status, pad_to,
                       smax, twist_mask);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    // persistent grids: what one order can keep resident (CUs x SIMDs x waves per SIMD), capped by the work there can be
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    // a unit holds at least one trajectory, so B units bound every order's list
    auto waves_for = [&](int per_simd) { const int64_t cap = (int64_t)cus * 4 * per_simd; return (int)(B < cap ? B : cap); };
    // CSP_MIXED_FORK=1: the orders' launches on forked streams (concurrent, but the event fork / join between hardware
    // queues costs tens of microseconds); default: one after the other on the caller's stream
    static const bool fork = [] { const char *e = std::getenv("CSP_MIXED_FORK"); return e && e[0] == '1'; }();
    Fork *f = fork ? fork_for_device() : nullptr;
    if (twist_mask) {
        int wgs = cus * 2;   // one wave per SIMD: two workgroups of two waves per CU
        const int slots = twist_slots(B);
        if (wgs > slots) wgs = slots;
        if ((e = twist::launch_twist(a, f32, perm, coef_off, tab, ckws, twist_slot_doubles_per_role(smax), wgs, st)) != hipSuccess) return e;
        if (twist_mask == 15 && smax <= twist::SMAX) return hipSuccess;   // nothing can be left for the chunked kernels
    }
    if (!f) {   // no side streams: the four orders one after the other on the caller's stream
        if ((e = launch_order<5>(a, f32, perm, coef_off, tab, waves_for(1), st)) != hipSuccess) return e;
        if ((e = launch_order<4>(a, f32, perm, coef_off, tab, waves_for(2), st)) != hipSuccess) return e;
        if ((e = launch_order<3>(a, f32, perm, coef_off, tab, waves_for(2), st)) != hipSuccess) return e;
        return launch_order<2>(a, f32, perm, coef_off, tab, waves_for(2), st);
    }
    if ((e = hipEventRecord(f->begin, st)) != hipSuccess) return e;
    for (int i = 0; i < 3; ++i)
        if ((e = hipStreamWaitEvent(f->s[i], f->begin, 0)) != hipSuccess) return e;
    // the most expensive order first; order 2 shares the caller's stream with order 5
    if ((e = launch_order<5>(a, f32, perm, coef_off, tab, waves_for(1), st)) != hipSuccess) return e;
    if ((e = launch_order<4>(a, f32, perm, coef_off, tab, waves_for(2), f->s[0])) != hipSuccess) return e;
    if ((e = launch_order<3>(a, f32, perm, coef_off, tab, waves_for(2), f->s[1])) != hipSuccess) return e;
    if ((e = launch_order<2>(a, f32, perm, coef_off, tab, waves_for(2), f->s[2])) != hipSuccess) return e;
    for (int i = 0; i < 3; ++i) {
        if ((e = hipEventRecord(f->done[i], f->s[i])) != hipSuccess) return e;
        if ((e = hipStreamWaitEvent(st, f->done[i], 0)) != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace csp
