// minsnap_capi.hip -- implementation of the C-ABI in include/csp_minsnap.h.
// Host-side validation, workspace carving, host<->device staging for CSP_MEM_HOST callers and
// kernel dispatch.  No CPU compute path exists: every entry point ends in a HIP launch or an
// error code.
#include "../../include/csp_minsnap.h"
#include "minsnap_launch.h"
#include "minsnap_hoststage.h"
#include "minsnap_timealloc.h"
#include "minsnap_shard_schedule.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

thread_local std::string g_last_hip_error;

int hip_fail(hipError_t e, const char *what) {
    g_last_hip_error = std::string(what) + ": " + hipGetErrorString(e);
    return CSP_ERR_HIP;
}
#define CSP_HIP(call)                                   \
    do {                                                \
        hipError_t e_ = (call);                         \
        if (e_ != hipSuccess) return hip_fail(e_, #call); \
    } while (0)

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct Shape {
    bool f32, ragged;
    int order, S, Smax;
    int64_t B;
    size_t elt;
};

int validate(const csp_minsnap_desc *d, Shape &s) {
    if (!d) return CSP_ERR_INVALID_ARG;
    if (d->abi_version != CSP_MINSNAP_ABI_VERSION) return CSP_ERR_INVALID_ARG;
    if (d->dtype != CSP_DTYPE_F64 && d->dtype != CSP_DTYPE_F32) return CSP_ERR_INVALID_ARG;
    if (d->mem_space != CSP_MEM_HOST && d->mem_space != CSP_MEM_DEVICE) return CSP_ERR_INVALID_ARG;
    if (d->order < 1) return CSP_ERR_INVALID_ARG;
    if (d->order > 5) return CSP_ERR_UNSUPPORTED;
    if (d->batch < 0 || d->num_segments < 0) return CSP_ERR_INVALID_ARG;
    if (d->path_weight < 0.0 || d->vel_zero_weight < 0.0) return CSP_ERR_INVALID_ARG;
    if ((d->flags & CSP_FLAG_SEGMENT_MAJOR) && d->num_segments == 0) return CSP_ERR_INVALID_ARG;
    s.f32 = d->dtype == CSP_DTYPE_F32;
    s.elt = s.f32 ? 4 : 8;
    s.order = d->order;
    s.B = d->batch;
    s.ragged = d->num_segments == 0;
    if (s.ragged) {
        if (!d->seg_offsets || d->max_segments < 1) return CSP_ERR_INVALID_ARG;
        s.S = 0;
        s.Smax = d->max_segments;
    } else {
        s.S = s.Smax = d->num_segments;
    }
    return CSP_OK;
}

bool use_fixed(const csp_minsnap_desc *d, const Shape &s) {
    if (d->flags & CSP_FLAG_FORCE_GENERIC) return false;
    return csp::fixed_supported(s.order, s.S, s.f32, d->path_weight, s.ragged, (d->flags & CSP_FLAG_SEGMENT_MAJOR) != 0);
}

// very long trajectories (256 < S <= 1024; from 17 segments at order 5 or with CSP_FLAG_SPAN): spans of 16 segments per
// lane (minsnap_span.hip).  Below 257 segments the chunked kernel is faster: the span kernel re-reads its
// inputs once per elimination step and 2048 resident waves x 34 KB do not stay in L2.
// csp_minsnap_solve_batch_sharded decides span-vs-chunked ONCE from the whole batch (the order-5 rule below looks at
// the batch size) and pins that choice for its per-device chunks, so that sharding never changes the arithmetic.
thread_local int g_span_override = -1;   // -1: decide here; 0 / 1: decided by the caller

bool use_span(const csp_minsnap_desc *d, const Shape &s) {
    if ((d->flags & CSP_FLAG_FORCE_GENERIC) || use_fixed(d, s)) return false;
    if (g_span_override >= 0)
        return g_span_override == 1 && csp::span_supported(s.order, s.Smax, s.f32 && (d->flags & CSP_FLAG_F32_ARITH), d->path_weight,
                                                           (d->flags & CSP_FLAG_SEGMENT_MAJOR) != 0);
    // order 5 (4x4 blocks) is the exception: the chunked kernel needs 392 registers there (one wave per
    // SIMD) and loses to the span kernel from 17 segments on (measured 1.3-1.75x at S = 20..128)
    // (with at least a wave per SIMD of span lanes: below that the chunked kernel's 4x more lanes win)
    const bool o5_big = s.order == 5 && s.Smax > 16 && (s.B << csp::span_lanes_log2(s.Smax)) >= 65536;
    if (s.Smax <= 256 && !(d->flags & CSP_FLAG_SPAN) && !o5_big) return false;
    return csp::span_supported(s.order, s.Smax, s.f32 && (d->flags & CSP_FLAG_F32_ARITH), d->path_weight,
                               (d->flags & CSP_FLAG_SEGMENT_MAJOR) != 0);
}

// the multi-lane workspace-free kernel takes what the fixed buckets and the span kernel do not: short
// ragged batches, fp32 storage at S <= 16 (minsnap_chunked.hip)
bool use_chunked(const csp_minsnap_desc *d, const Shape &s) {
    if ((d->flags & CSP_FLAG_FORCE_GENERIC) || use_fixed(d, s) || use_span(d, s)) return false;
    return csp::chunked_supported(s.order, s.Smax, s.f32 && (d->flags & CSP_FLAG_F32_ARITH), d->path_weight,
                                  (d->flags & CSP_FLAG_SEGMENT_MAJOR) != 0);
}

size_t ws_bytes(const csp_minsnap_desc *d, const Shape &s, size_t *tstar_off) {
    if (use_fixed(d, s) || use_span(d, s) || use_chunked(d, s)) { if (tstar_off) *tstar_off = 0; return 0; }
    const size_t ws_elt = (s.f32 && (d->flags & CSP_FLAG_F32_ARITH)) ? 4 : 8;  // workspace holds the arithmetic type
    size_t factors = align_up((size_t)(s.Smax > 1 ? s.Smax - 1 : 0) * csp::generic_ws_entries(s.order) *
                                  (size_t)s.B * ws_elt, 256);
    if (tstar_off) *tstar_off = factors;
    size_t ts = d->path_weight > 0.0 ? align_up((size_t)s.Smax * (size_t)s.B * sizeof(int), 256) : 0;
    return factors + ts;
}

int select_device(int device_id) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return CSP_ERR_NO_DEVICE;
    if (device_id >= n) return CSP_ERR_INVALID_ARG;
    if (device_id >= 0) CSP_HIP(hipSetDevice(device_id));
    int cur = 0;
    CSP_HIP(hipGetDevice(&cur));
    // the architecture check costs a property query: once per device and thread, not per call
    static thread_local unsigned long long checked_ok = 0;
    if (cur < 64 && ((checked_ok >> cur) & 1ull)) return CSP_OK;
    hipDeviceProp_t p;
    CSP_HIP(hipGetDeviceProperties(&p, cur));
    if (std::strncmp(p.gcnArchName, "gfx950", 6) != 0) {
        g_last_hip_error = std::string("device is ") + p.gcnArchName + ", this library is built for gfx950 only";
        return CSP_ERR_NO_DEVICE;
    }
    if (cur < 64) checked_ok |= 1ull << cur;
    return CSP_OK;
}

int current_device() {
    int cur = 0;
    (void)hipGetDevice(&cur);
    return cur;
}

// Smallest double x with sqrt(x) >= d under IEEE rounding (sqrt is monotone), so that the kernels can
// test the squared distance: d2 >= x  <=>  sqrt(d2) >= d.  d <= 0 keeps everything (x = 0), NaN nothing.
double keep_threshold(double d) {
    if (std::isnan(d)) return d;
    if (d <= 0.0) return 0.0;
    if (std::isinf(d)) return d;
    double x = d * d;
    if (std::isinf(x)) x = std::numeric_limits<double>::max();
    while (x > 0.0 && std::sqrt(std::nextafter(x, 0.0)) >= d) x = std::nextafter(x, 0.0);
    while (std::sqrt(x) < d) x = std::nextafter(x, std::numeric_limits<double>::infinity());
    return x;
}

int dispatch(const csp_minsnap_desc *d, const Shape &s, const void *wp, const void *tm, const void *bc,
             void *co, double *max_dev, int32_t *status, const int64_t *seg_off, const double *vw_per,
             void *ws, size_t ws_size, hipStream_t st, const int32_t *skip = nullptr, int *tau_buf = nullptr, int tau_mode = 0) {
    size_t tstar_off = 0;
    const size_t need = ws_bytes(d, s, &tstar_off);
    if (need > 0 && (!ws || ws_size < need)) return CSP_ERR_WORKSPACE;
    csp::GenericArgs a;
    a.wp = wp; a.times = tm; a.bc = bc; a.coeffs = co; a.max_dev = max_dev; a.status = status;
    a.seg_off = s.ragged ? seg_off : nullptr;
    a.ws = ws;
    a.tstar = d->path_weight > 0.0 ? (int *)((char *)ws + tstar_off) : nullptr;
    a.vw_per = vw_per;
    a.path_weight = d->path_weight;
    a.vel_zero_weight = d->vel_zero_weight;
    a.B = s.B; a.S = s.S; a.order = s.order; a.bc_per_traj = d->bc_per_trajectory ? 1 : 0;
    a.seg_major = (d->flags & CSP_FLAG_SEGMENT_MAJOR) ? 1 : 0;
    a.Btotal = s.B;
    a.Boffset = 0;
    a.persistent = (d->flags & CSP_FLAG_NO_PERSISTENT) ? 0 : 1;
    a.skip = skip;
    a.tau_mode = 0;
    // inside the re-solve loop: the path kernel keeps its t* indices in tau_buf, the generic kernel in its workspace
    if (use_fixed(d, s)) { if (tau_buf) { a.tstar = tau_buf; a.tau_mode = tau_mode; } }
    else a.tau_mode = tau_mode;
    // the fixed kernel moves 16-byte pieces (LDS-DMA, ds_read_b128, dwordx4 stores)
    const bool aligned = (((uintptr_t)wp | (uintptr_t)tm | (uintptr_t)co) & 15u) == 0;
    if (use_fixed(d, s) && !aligned) return CSP_ERR_INVALID_ARG;
    // the chunked kernel reads scalars and stores 16-byte pieces (8-byte for fp32 with odd order)
    if ((use_chunked(d, s) || use_span(d, s)) && ((uintptr_t)co & ((s.f32 && (s.order & 1)) ? 7u : 15u))) return CSP_ERR_INVALID_ARG;
    hipError_t e = use_fixed(d, s) ? csp::launch_fixed(a, st)
                 : use_span(d, s)    ? csp::launch_span(a, s.f32, s.Smax, st)
                 : use_chunked(d, s) ? csp::launch_chunked(a, s.f32, s.Smax, st)
                                     : csp::launch_generic(a, s.f32, (d->flags & CSP_FLAG_F32_ARITH) != 0, st);
    if (e != hipSuccess) return hip_fail(e, "kernel launch");
    return CSP_OK;
}


// ---- csp_minsnap_solve_batch_sharded for a batch RESIDENT ON A ROOT DEVICE: scatter / solve / gather over RCCL -------------
// (the schedule itself is minsnap_shard_schedule.h; this is its transport).  One process, one thread: single-process
// communicators (ncclCommInitAll, cached per device list), per device a compute stream and a communication stream, an arena
// on every peer for its shard's inputs and outputs.  UNVERIFIED ON MORE THAN ONE GPU: the development boxes have one; what
// runs there is the ngpu = 1 degenerate path (no communicator) and, on the CPU, the schedule over a recording transport.
struct ShardNode {
    std::vector<int> devs;              // HIP ordinals, devs[root] owns the caller's buffers
    std::vector<ncclComm_t> comms;      // empty for one device
    std::vector<hipStream_t> compute, comm;
};

int nccl_fail(ncclResult_t r, const char *what) {
    g_last_hip_error = std::string(what) + ": " + ncclGetErrorString(r);
    return CSP_ERR_HIP;
}
#define CSP_NCCL(call)                                       \
    do {                                                     \
        ncclResult_t r_ = (call);                            \
        if (r_ != ncclSuccess) return nccl_fail(r_, #call);  \
    } while (0)

// cached for the life of the process (communicator set-up costs hundreds of milliseconds)
int shard_node(const std::vector<int> &devs, ShardNode **out) {
    static std::mutex m;
    static std::map<std::vector<int>, ShardNode *> cache;
    std::lock_guard<std::mutex> g(m);
    auto it = cache.find(devs);
    if (it != cache.end()) { *out = it->second; return CSP_OK; }
    std::unique_ptr<ShardNode> n(new ShardNode());
    n->devs = devs;
    n->compute.resize(devs.size());
    n->comm.resize(devs.size());
    for (size_t i = 0; i < devs.size(); ++i) {
        CSP_HIP(hipSetDevice(devs[i]));
        CSP_HIP(hipStreamCreateWithFlags(&n->compute[i], hipStreamNonBlocking));
        CSP_HIP(hipStreamCreateWithFlags(&n->comm[i], hipStreamNonBlocking));
    }
    if (devs.size() > 1) {
        n->comms.resize(devs.size());
        CSP_NCCL(ncclCommInitAll(n->comms.data(), (int)devs.size(), devs.data()));
    }
    *out = n.get();
    cache[devs] = n.release();
    return CSP_OK;
}

struct RcclTransport {
    const csp_minsnap_desc *desc;
    Shape s;
    ShardNode *node;
    int root, nchunks;
    // caller's buffers on the root device
    const char *wp, *tm, *bc;
    char *co;
    double *max_dev;
    int32_t *status;
    // per device: arena + carved offsets of its shard (peers only), events
    struct Dev {
        csp::Arena *arena = nullptr;
        int64_t lo = 0, hi = 0;
        size_t o_wp = 0, o_tm = 0, o_bc = 0, o_vw = 0, o_co = 0, o_md = 0, o_st = 0, o_ws = 0, ws_bytes = 0;
        std::vector<hipEvent_t> in_ready, solved;
    };
    std::vector<Dev> dev;
    size_t m() const { return 2 * (size_t)s.order; }
    size_t wp_bytes(int64_t n) const { return (size_t)n * (size_t)(s.S + 1) * 3 * s.elt; }
    size_t tm_bytes(int64_t n) const { return (size_t)n * (size_t)s.S * s.elt; }
    size_t co_bytes(int64_t n) const { return (size_t)n * (size_t)s.S * 3 * m() * s.elt; }

    int setup() {
        const int nd = (int)node->devs.size();
        dev.resize((size_t)nd);
        for (int g = 0; g < nd; ++g) {
            Dev &d = dev[(size_t)g];
            csp::shard::shard_range(s.B, nd, g, d.lo, d.hi);
            const int64_t n = d.hi - d.lo;
            CSP_HIP(hipSetDevice(node->devs[(size_t)g]));
            d.in_ready.resize((size_t)nchunks);
            d.solved.resize((size_t)nchunks);
            for (int c = 0; c < nchunks; ++c) {
                CSP_HIP(hipEventCreateWithFlags(&d.in_ready[(size_t)c], hipEventDisableTiming));
                CSP_HIP(hipEventCreateWithFlags(&d.solved[(size_t)c], hipEventDisableTiming));
            }
            // the solve of a piece may need the generic kernel's workspace: sized for the largest piece
            csp_minsnap_desc pd = *desc;
            pd.batch = (n + nchunks - 1) / nchunks;
            Shape ps;
            if (validate(&pd, ps) != CSP_OK) return CSP_ERR_INVALID_ARG;
            d.ws_bytes = ws_bytes(&pd, ps, nullptr);
            if (g == root && d.ws_bytes == 0) continue;
            d.arena = csp::arena_acquire(node->devs[(size_t)g]);
            size_t top = 0;
            auto carve = [&](size_t bytes) { const size_t o = top; top = align_up(top + bytes, 256); return o; };
            if (g != root) {
                d.o_wp = carve(wp_bytes(n));
                d.o_tm = carve(tm_bytes(n));
                d.o_bc = carve(desc->bc_per_trajectory ? (size_t)n * 12 * s.elt : 12 * s.elt);
                d.o_vw = carve(desc->vel_zero_weight_per_traj ? (size_t)n * 8 : 0);
                d.o_co = carve(co_bytes(n));
                d.o_md = carve(max_dev ? (size_t)n * 8 : 0);
                d.o_st = carve(status ? (size_t)n * 4 : 0);
            }
            d.o_ws = carve(d.ws_bytes);
            CSP_HIP(d.arena->reserve(top));
        }
        return CSP_OK;
    }
    void teardown() {
        for (size_t g = 0; g < dev.size(); ++g) {
            if (hipSetDevice(node->devs[g]) != hipSuccess) continue;
            for (hipEvent_t e : dev[g].in_ready) if (e) (void)hipEventDestroy(e);
            for (hipEvent_t e : dev[g].solved) if (e) (void)hipEventDestroy(e);
            if (dev[g].arena) csp::arena_release(dev[g].arena);
        }
        (void)hipSetDevice(node->devs[(size_t)root]);
    }

    // ---- the Transport concept of minsnap_shard_schedule.h ----
    int scatter_begin(int) { if (!node->comms.empty()) CSP_NCCL(ncclGroupStart()); return 0; }
    int scatter_piece(const csp::shard::Piece &p) {
        Dev &d = dev[(size_t)p.dev];
        const int64_t n = p.hi - p.lo, off = p.lo - d.lo;
        hipStream_t rs = node->comm[(size_t)root], ps = node->comm[(size_t)p.dev];
        ncclComm_t rc = node->comms[(size_t)root], pc = node->comms[(size_t)p.dev];
        CSP_NCCL(ncclSend(wp + wp_bytes(p.lo), wp_bytes(n), ncclChar, p.dev, rc, rs));
        CSP_NCCL(ncclRecv(d.arena->dev + d.o_wp + wp_bytes(off), wp_bytes(n), ncclChar, root, pc, ps));
        CSP_NCCL(ncclSend(tm + tm_bytes(p.lo), tm_bytes(n), ncclChar, p.dev, rc, rs));
        CSP_NCCL(ncclRecv(d.arena->dev + d.o_tm + tm_bytes(off), tm_bytes(n), ncclChar, root, pc, ps));
        if (desc->bc_per_trajectory) {
            CSP_NCCL(ncclSend(bc + (size_t)p.lo * 12 * s.elt, (size_t)n * 12 * s.elt, ncclChar, p.dev, rc, rs));
            CSP_NCCL(ncclRecv(d.arena->dev + d.o_bc + (size_t)off * 12 * s.elt, (size_t)n * 12 * s.elt, ncclChar, root, pc, ps));
        } else if (p.chunk == 0) {
            CSP_NCCL(ncclSend(bc, 12 * s.elt, ncclChar, p.dev, rc, rs));
            CSP_NCCL(ncclRecv(d.arena->dev + d.o_bc, 12 * s.elt, ncclChar, root, pc, ps));
        }
        if (desc->vel_zero_weight_per_traj) {
            CSP_NCCL(ncclSend(desc->vel_zero_weight_per_traj + p.lo, (size_t)n * 8, ncclChar, p.dev, rc, rs));
            CSP_NCCL(ncclRecv(d.arena->dev + d.o_vw + (size_t)off * 8, (size_t)n * 8, ncclChar, root, pc, ps));
        }
        return 0;
    }
    int scatter_end(int c) {
        if (node->comms.empty()) return 0;
        CSP_NCCL(ncclGroupEnd());
        for (size_t g = 0; g < dev.size(); ++g) {
            if ((int)g == root) continue;
            CSP_HIP(hipSetDevice(node->devs[g]));
            CSP_HIP(hipEventRecord(dev[g].in_ready[(size_t)c], node->comm[g]));
        }
        return 0;
    }
    int solve(const csp::shard::Piece &p) {
        Dev &d = dev[(size_t)p.dev];
        const int64_t n = p.hi - p.lo, off = p.lo - d.lo;
        CSP_HIP(hipSetDevice(node->devs[(size_t)p.dev]));
        hipStream_t st = node->compute[(size_t)p.dev];
        csp_minsnap_desc pd = *desc;
        pd.batch = n;
        pd.mem_space = CSP_MEM_DEVICE;
        pd.device_id = node->devs[(size_t)p.dev];
        Shape ps;
        int rc = validate(&pd, ps);
        if (rc != CSP_OK) return rc;
        void *ws = d.ws_bytes ? (void *)(d.arena->dev + d.o_ws) : nullptr;
        if (p.dev == root) {
            rc = dispatch(&pd, ps, wp + wp_bytes(p.lo), tm + tm_bytes(p.lo), bc + (desc->bc_per_trajectory ? (size_t)p.lo * 12 * s.elt : 0),
                          co + co_bytes(p.lo), max_dev ? max_dev + p.lo : nullptr, status ? status + p.lo : nullptr, nullptr,
                          desc->vel_zero_weight_per_traj ? desc->vel_zero_weight_per_traj + p.lo : nullptr, ws, d.ws_bytes, st);
        } else {
            CSP_HIP(hipStreamWaitEvent(st, d.in_ready[(size_t)p.chunk], 0));
            char *a = d.arena->dev;
            rc = dispatch(&pd, ps, a + d.o_wp + wp_bytes(off), a + d.o_tm + tm_bytes(off),
                          a + d.o_bc + (desc->bc_per_trajectory ? (size_t)off * 12 * s.elt : 0), a + d.o_co + co_bytes(off),
                          max_dev ? (double *)(a + d.o_md) + off : nullptr, status ? (int32_t *)(a + d.o_st) + off : nullptr, nullptr,
                          desc->vel_zero_weight_per_traj ? (const double *)(a + d.o_vw) + off : nullptr, ws, d.ws_bytes, st);
        }
        if (rc != CSP_OK) return rc;
        CSP_HIP(hipEventRecord(d.solved[(size_t)p.chunk], st));
        return 0;
    }
    int gather_begin(int c) {
        if (node->comms.empty()) return 0;
        // a peer's communication stream sends piece c after ITS solve of c only
        for (size_t g = 0; g < dev.size(); ++g) {
            if ((int)g == root) continue;
            const csp::shard::Piece p = csp::shard::piece_of(s.B, (int)dev.size(), nchunks, (int)g, c);
            if (p.hi <= p.lo) continue;
            CSP_HIP(hipSetDevice(node->devs[g]));
            CSP_HIP(hipStreamWaitEvent(node->comm[g], dev[g].solved[(size_t)c], 0));
        }
        CSP_NCCL(ncclGroupStart());
        return 0;
    }
    int gather_piece(const csp::shard::Piece &p) {
        Dev &d = dev[(size_t)p.dev];
        const int64_t n = p.hi - p.lo, off = p.lo - d.lo;
        hipStream_t rs = node->comm[(size_t)root], ps = node->comm[(size_t)p.dev];
        ncclComm_t rc = node->comms[(size_t)root], pc = node->comms[(size_t)p.dev];
        char *a = d.arena->dev;
        CSP_NCCL(ncclSend(a + d.o_co + co_bytes(off), co_bytes(n), ncclChar, root, pc, ps));
        CSP_NCCL(ncclRecv(co + co_bytes(p.lo), co_bytes(n), ncclChar, p.dev, rc, rs));
        if (max_dev) {
            CSP_NCCL(ncclSend(a + d.o_md + (size_t)off * 8, (size_t)n * 8, ncclChar, root, pc, ps));
            CSP_NCCL(ncclRecv(max_dev + p.lo, (size_t)n * 8, ncclChar, p.dev, rc, rs));
        }
        if (status) {
            CSP_NCCL(ncclSend(a + d.o_st + (size_t)off * 4, (size_t)n * 4, ncclChar, root, pc, ps));
            CSP_NCCL(ncclRecv(status + p.lo, (size_t)n * 4, ncclChar, p.dev, rc, rs));
        }
        return 0;
    }
    int gather_end(int) { if (!node->comms.empty()) CSP_NCCL(ncclGroupEnd()); return 0; }
    int finish() {
        for (size_t g = 0; g < dev.size(); ++g) {
            CSP_HIP(hipSetDevice(node->devs[g]));
            CSP_HIP(hipStreamSynchronize(node->compute[g]));
            CSP_HIP(hipStreamSynchronize(node->comm[g]));
        }
        return 0;
    }
};

// chunks per shard of the device-resident sharded call: the gather of chunk i overlaps the solve of chunk i + 1
// (CSP_SHARD_CHUNKS overrides; a chunk keeps at least 4096 trajectories)
int shard_chunks(int64_t per_dev) {
    static const int forced = [] { const char *e = std::getenv("CSP_SHARD_CHUNKS"); return e ? std::atoi(e) : 0; }();
    int n = forced > 0 ? forced : 4;
    while (n > 1 && per_dev / n < 4096) --n;
    return n;
}

int solve_sharded_device(const csp_minsnap_desc *desc, const Shape &s, const void *waypoints, const void *times, const void *bc,
                         void *coeffs, double *max_dev, int32_t *status, const std::vector<int> &ordinals, int ngpu) {
    if (s.ragged) return CSP_ERR_UNSUPPORTED;   // the root-resident form shards uniform batches (ragged ones: host-memory form)
    // the root is the device that owns the caller's buffers: desc->device_id, or the current device
    int root_dev = desc->device_id;
    if (root_dev < 0) CSP_HIP(hipGetDevice(&root_dev));
    std::vector<int> devs;
    devs.push_back(root_dev);
    for (int o : ordinals)
        if (o != root_dev && (int)devs.size() < ngpu) devs.push_back(o);
    if ((int)devs.size() != ngpu) return CSP_ERR_INVALID_ARG;
    bool root_ok = false;
    for (int o : ordinals) root_ok |= o == root_dev;
    if (!root_ok) return CSP_ERR_NO_DEVICE;
    ShardNode *node = nullptr;
    int rc = shard_node(devs, &node);
    if (rc != CSP_OK) return rc;
    CSP_HIP(hipSetDevice(root_dev));
    CSP_HIP(hipDeviceSynchronize());   // the caller's inputs are complete (the entry takes no stream)
    RcclTransport t;
    t.desc = desc; t.s = s; t.node = node; t.root = 0;
    t.nchunks = shard_chunks(s.B / ngpu);
    t.wp = (const char *)waypoints; t.tm = (const char *)times; t.bc = (const char *)bc;
    t.co = (char *)coeffs; t.max_dev = max_dev; t.status = status;
    const int span_before = g_span_override;
    g_span_override = use_span(desc, s) ? 1 : 0;   // decided from the WHOLE batch, pinned for the pieces
    rc = t.setup();
    if (rc == CSP_OK) rc = csp::shard::run(t, s.B, ngpu, 0, t.nchunks);
    if (rc != CSP_OK) (void)t.finish();
    t.teardown();
    g_span_override = span_before;
    return rc;
}

}  // namespace

extern "C" {

const char *csp_minsnap_version(void) { return "csp-minsnap 0.1.0 (gfx950, abi 1)"; }

const char *csp_minsnap_strerror(int st) {
    switch (st) {
        case CSP_OK: return "ok";
        case CSP_ERR_INVALID_ARG: return "invalid argument";
        case CSP_ERR_UNSUPPORTED: return "unsupported order (reference int arithmetic overflows from order 6)";
        case CSP_ERR_WORKSPACE: return "workspace missing or too small";
        case CSP_ERR_HIP: return "HIP runtime error";
        case CSP_ERR_NO_DEVICE: return "no gfx950 device (no CPU fallback exists)";
    }
    return "unknown status";
}

const char *csp_minsnap_last_hip_error(void) { return g_last_hip_error.c_str(); }

int csp_minsnap_device_count(void) {
    int n = 0, good = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    for (int i = 0; i < n; ++i) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, i) == hipSuccess && std::strncmp(p.gcnArchName, "gfx950", 6) == 0) ++good;
    }
    return good;
}

size_t csp_minsnap_workspace_bytes(const csp_minsnap_desc *desc) {
    Shape s;
    if (validate(desc, s) != CSP_OK) return 0;
    return ws_bytes(desc, s, nullptr);
}

const char *csp_minsnap_kernel_name(const csp_minsnap_desc *desc) {
    static thread_local char name[64];
    Shape s;
    if (validate(desc, s) != CSP_OK) return nullptr;
    if (use_fixed(desc, s)) return csp::fixed_kernel_name(s.order, s.S, desc->path_weight > 0.0);
    if (use_span(desc, s)) {
        std::snprintf(name, sizeof name, "span_o%d_%s_l%d%s", s.order, s.f32 ? "f32io_f64" : "f64",
                      1 << csp::span_lanes_log2(s.Smax), s.ragged ? "_ragged" : "");
        return name;
    }
    if (use_chunked(desc, s)) {
        std::snprintf(name, sizeof name, "chunked_o%d_%s_l%d%s", s.order, s.f32 ? "f32io_f64" : "f64",
                      1 << csp::chunked_lanes_log2(s.Smax), s.ragged ? "_ragged" : "");
        return name;
    }
    std::snprintf(name, sizeof name, "generic_o%d_%s%s", s.order,
                  !s.f32 ? "f64" : ((desc->flags & CSP_FLAG_F32_ARITH) ? "f32" : "f32io_f64"), s.ragged ? "_ragged" : "");
    return name;
}

int csp_minsnap_solve_batch(const csp_minsnap_desc *desc, const void *waypoints, const void *times,
                            const void *bc, void *coeffs, double *max_dev, int32_t *status,
                            void *workspace, size_t workspace_bytes, void *hip_stream) {
    Shape s;
    int rc = validate(desc, s);
    if (rc != CSP_OK) return rc;
    if (s.B == 0) return CSP_OK;
    if (!waypoints || !times || !bc || !coeffs) return CSP_ERR_INVALID_ARG;
    rc = select_device(desc->device_id);
    if (rc != CSP_OK) return rc;
    hipStream_t st = (hipStream_t)hip_stream;

    if (desc->mem_space == CSP_MEM_DEVICE)
        return dispatch(desc, s, waypoints, times, bc, coeffs, max_dev, status, desc->seg_offsets,
                        desc->vel_zero_weight_per_traj, workspace, workspace_bytes, st);

    // CSP_MEM_HOST: stage through a cached per-device arena (minsnap_hoststage.h), synchronously.
    int64_t total_seg;
    if (s.ragged) {
        total_seg = desc->seg_offsets[s.B];
        for (int64_t b = 0; b < s.B; ++b) {
            const int64_t n = desc->seg_offsets[b + 1] - desc->seg_offsets[b];
            if (n < 0 || n > s.Smax) return CSP_ERR_INVALID_ARG;
        }
    } else {
        total_seg = s.B * (int64_t)s.S;
    }
    const size_t m = 2 * (size_t)s.order;
    const size_t n_wp = (size_t)(total_seg + s.B) * 3 * s.elt, n_tm = (size_t)total_seg * s.elt;
    const size_t n_bc = (size_t)(desc->bc_per_trajectory ? s.B : 1) * 12 * s.elt;
    const size_t n_co = (size_t)total_seg * 3 * m * s.elt;
    const size_t n_ws = ws_bytes(desc, s, nullptr);
    csp::HostCall hc(current_device(), st);
    const size_t o_wp = hc.in(waypoints, n_wp), o_tm = hc.in(times, n_tm), o_bc = hc.in(bc, n_bc);
    const size_t o_so = s.ragged ? hc.in(desc->seg_offsets, (size_t)(s.B + 1) * 8) : 0;
    const size_t o_vw = desc->vel_zero_weight_per_traj ? hc.in(desc->vel_zero_weight_per_traj, (size_t)s.B * 8) : 0;
    const size_t o_co = hc.out(coeffs, n_co);
    const size_t o_md = max_dev ? hc.out(max_dev, (size_t)s.B * 8) : 0, o_st = status ? hc.out(status, (size_t)s.B * 4) : 0;
    const size_t o_ws = hc.scratch(n_ws);
    CSP_HIP(hc.upload());
    rc = dispatch(desc, s, hc.ptr(o_wp), hc.ptr(o_tm), hc.ptr(o_bc), hc.ptr(o_co), max_dev ? hc.ptr<double>(o_md) : nullptr,
                  status ? hc.ptr<int32_t>(o_st) : nullptr, s.ragged ? hc.ptr<const int64_t>(o_so) : nullptr,
                  desc->vel_zero_weight_per_traj ? hc.ptr<const double>(o_vw) : nullptr, hc.ptr(o_ws), n_ws, st);
    if (rc != CSP_OK) return rc;
    CSP_HIP(hc.download());
    return CSP_OK;
}

int csp_minsnap_solve_multi(const csp_minsnap_desc *desc, int n, const int64_t *batches, const void *const *waypoints,
                            const void *const *times, const void *const *bc, void *const *coeffs, int32_t *const *status,
                            void *hip_stream) {
    if (!desc || n < 0) return CSP_ERR_INVALID_ARG;
    if (n == 0) return CSP_OK;
    if (!batches || !waypoints || !times || !bc || !coeffs) return CSP_ERR_INVALID_ARG;
    csp_minsnap_desc d = *desc;
    int64_t total = 0;
    for (int k = 0; k < n; ++k) {
        if (batches[k] < 0) return CSP_ERR_INVALID_ARG;
        if (batches[k] && (!waypoints[k] || !times[k] || !bc[k] || !coeffs[k] || (status && !status[k]))) return CSP_ERR_INVALID_ARG;
        total += batches[k];
    }
    d.batch = total;
    Shape s;
    int rc = validate(&d, s);
    if (rc != CSP_OK) return rc;
    if (d.mem_space != CSP_MEM_DEVICE || s.ragged || d.vel_zero_weight_per_traj) return CSP_ERR_UNSUPPORTED;
    if (total == 0) return CSP_OK;
    rc = select_device(d.device_id);
    if (rc != CSP_OK) return rc;
    hipStream_t st = (hipStream_t)hip_stream;
    const bool one_launch = use_fixed(&d, s) && d.path_weight == 0.0 && !(d.flags & CSP_FLAG_SEGMENT_MAJOR);
    if (!one_launch) {
        // shapes the table-driven kernels do not serve: one ordinary launch per batch (still one C-ABI call), provided no
        // workspace is needed
        for (int k = 0; k < n; ++k) {
            if (!batches[k]) continue;
            csp_minsnap_desc dk = d;
            dk.batch = batches[k];
            Shape sk;
            if ((rc = validate(&dk, sk)) != CSP_OK) return rc;
            if (ws_bytes(&dk, sk, nullptr) != 0) return CSP_ERR_UNSUPPORTED;
            rc = dispatch(&dk, sk, waypoints[k], times[k], bc[k], coeffs[k], nullptr, status ? status[k] : nullptr, nullptr, nullptr, nullptr, 0, st);
            if (rc != CSP_OK) return rc;
        }
        return CSP_OK;
    }
    csp::GenericArgs a;
    a.wp = a.times = a.bc = nullptr; a.coeffs = nullptr; a.max_dev = nullptr; a.status = nullptr; a.seg_off = nullptr; a.ws = nullptr;
    a.tstar = nullptr; a.vw_per = nullptr;
    a.path_weight = 0.0; a.vel_zero_weight = d.vel_zero_weight;
    a.B = 0; a.S = s.S; a.order = s.order; a.bc_per_traj = d.bc_per_trajectory ? 1 : 0;
    a.seg_major = 0; a.Btotal = 0; a.Boffset = 0; a.persistent = 0; a.skip = nullptr; a.tau_mode = 0;
    for (int k0 = 0; k0 < n; k0 += 32) {
        csp::MultiTable mt;
        int slices = 0;
        mt.n = 0;
        for (int k = k0; k < n && k < k0 + 32; ++k) {
            if (!batches[k]) continue;
            if ((((uintptr_t)waypoints[k] | (uintptr_t)times[k] | (uintptr_t)coeffs[k]) & 15u) != 0) return CSP_ERR_INVALID_ARG;
            csp::MultiEntry &e = mt.e[mt.n];
            e.wp = waypoints[k]; e.tm = times[k]; e.bc = bc[k]; e.co = coeffs[k]; e.status = status ? status[k] : nullptr; e.B = batches[k];
            mt.first_slice[mt.n] = slices;
            slices += (int)((batches[k] + 63) / 64);
            ++mt.n;
        }
        if (!mt.n) continue;
        mt.first_slice[mt.n] = slices;
        a.multi = &mt;
        hipError_t e = csp::launch_fixed(a, st);
        if (e != hipSuccess) return hip_fail(e, "multi launch");
    }
    return CSP_OK;
}

size_t csp_minsnap_mixed_workspace_bytes(const csp_minsnap_desc *desc) {
    if (!desc || desc->batch < 0) return 0;
    return csp::mixed_workspace_bytes(desc->batch, desc->max_segments);
}

int csp_minsnap_solve_mixed(const csp_minsnap_desc *desc, const int32_t *orders, const void *waypoints, const void *times,
                            const void *bc, void *coeffs, int64_t *coeff_offsets_out, int32_t *status,
                            void *workspace, size_t workspace_bytes, void *hip_stream) {
    if (!desc || desc->abi_version != CSP_MINSNAP_ABI_VERSION) return CSP_ERR_INVALID_ARG;
    if (desc->dtype != CSP_DTYPE_F64 && desc->dtype != CSP_DTYPE_F32) return CSP_ERR_INVALID_ARG;
    if (desc->mem_space != CSP_MEM_HOST && desc->mem_space != CSP_MEM_DEVICE) return CSP_ERR_INVALID_ARG;
    if (desc->batch < 0 || desc->vel_zero_weight < 0.0 || desc->path_weight < 0.0) return CSP_ERR_INVALID_ARG;
    if (desc->path_weight != 0.0 || (desc->flags & (CSP_FLAG_F32_ARITH | CSP_FLAG_SEGMENT_MAJOR))) return CSP_ERR_UNSUPPORTED;
    if (desc->batch == 0) return CSP_OK;
    if (desc->batch > 0x7fffffff) return CSP_ERR_INVALID_ARG;   // trajectory indices are int32 on the device
    if (!desc->seg_offsets || desc->max_segments < 1 || desc->max_segments > 256) return CSP_ERR_INVALID_ARG;
    if (!orders || !waypoints || !times || !bc || !coeffs) return CSP_ERR_INVALID_ARG;
    int rc = select_device(desc->device_id);
    if (rc != CSP_OK) return rc;
    hipStream_t st = (hipStream_t)hip_stream;
    const bool f32 = desc->dtype == CSP_DTYPE_F32;
    const size_t elt = f32 ? 4 : 8;
    const int64_t B = desc->batch;
    const size_t need = csp::mixed_workspace_bytes(B, desc->max_segments);
    csp::GenericArgs a;
    a.max_dev = nullptr; a.ws = nullptr; a.tstar = nullptr;
    a.path_weight = 0.0;
    a.vel_zero_weight = desc->vel_zero_weight;
    a.B = B; a.S = desc->max_segments; a.order = 0; a.bc_per_traj = desc->bc_per_trajectory ? 1 : 0;
    a.seg_major = 0; a.Btotal = B; a.Boffset = 0; a.persistent = 1; a.skip = nullptr; a.tau_mode = 0;
    if (desc->mem_space == CSP_MEM_DEVICE) {
        if (!workspace || workspace_bytes < need) return CSP_ERR_WORKSPACE;
        if (((uintptr_t)coeffs & 15u) || ((uintptr_t)workspace & 15u)) return CSP_ERR_INVALID_ARG;
        a.wp = waypoints; a.times = times; a.bc = bc; a.coeffs = coeffs; a.status = status;
        a.seg_off = desc->seg_offsets; a.vw_per = desc->vel_zero_weight_per_traj;
        hipError_t e = csp::launch_mixed(a, f32, orders, workspace, coeff_offsets_out, st);
        return e == hipSuccess ? CSP_OK : hip_fail(e, "mixed launch");
    }
    // CSP_MEM_HOST: sizes from the caller's host arrays, staging through the cached arena, synchronous
    const int64_t total_seg = desc->seg_offsets[B];
    size_t total_co = 0;
    for (int64_t b = 0; b < B; ++b) {
        const int64_t n = desc->seg_offsets[b + 1] - desc->seg_offsets[b];
        if (n < 0 || n > desc->max_segments) return CSP_ERR_INVALID_ARG;
        if (orders[b] >= 1 && n > 0) { const size_t e = (size_t)n * 6 * (size_t)orders[b], pad = f32 ? 4 : 2; total_co += (e + pad - 1) / pad * pad; }
    }
    csp::HostCall hc(current_device(), st);
    const size_t o_wp = hc.in(waypoints, (size_t)(total_seg + B) * 3 * elt), o_tm = hc.in(times, (size_t)total_seg * elt);
    const size_t o_bc = hc.in(bc, (size_t)(desc->bc_per_trajectory ? B : 1) * 12 * elt);
    const size_t o_so = hc.in(desc->seg_offsets, (size_t)(B + 1) * 8), o_or = hc.in(orders, (size_t)B * 4);
    const size_t o_vw = desc->vel_zero_weight_per_traj ? hc.in(desc->vel_zero_weight_per_traj, (size_t)B * 8) : 0;
    const size_t o_co = hc.out(coeffs, total_co * elt);
    const size_t o_cf = coeff_offsets_out ? hc.out(coeff_offsets_out, (size_t)(B + 1) * 8) : 0;
    const size_t o_st = status ? hc.out(status, (size_t)B * 4) : 0;
    const size_t o_ws = hc.scratch(need);
    CSP_HIP(hc.upload());
    a.wp = hc.ptr(o_wp); a.times = hc.ptr(o_tm); a.bc = hc.ptr(o_bc); a.coeffs = hc.ptr(o_co);
    a.status = status ? hc.ptr<int32_t>(o_st) : nullptr;
    a.seg_off = hc.ptr<const int64_t>(o_so);
    a.vw_per = desc->vel_zero_weight_per_traj ? hc.ptr<const double>(o_vw) : nullptr;
    // skipped trajectories leave their block untouched: the caller's bytes must survive the round trip
    CSP_HIP(hipMemsetAsync(hc.ptr(o_co), 0, total_co * elt, st));
    hipError_t e = csp::launch_mixed(a, f32, hc.ptr<const int32_t>(o_or), hc.ptr(o_ws), coeff_offsets_out ? hc.ptr<int64_t>(o_cf) : nullptr, st);
    if (e != hipSuccess) return hip_fail(e, "mixed launch");
    CSP_HIP(hc.download());
    return CSP_OK;
}

int csp_minsnap_solve_batch_sharded(const csp_minsnap_desc *desc, const void *waypoints, const void *times,
                                    const void *bc, void *coeffs, double *max_dev, int32_t *status, int ngpu) {
    Shape s;
    int rc = validate(desc, s);
    if (rc != CSP_OK) return rc;
    if (desc->flags & CSP_FLAG_SEGMENT_MAJOR) return CSP_ERR_INVALID_ARG;
    if (s.B == 0) return CSP_OK;
    if (!waypoints || !times || !bc || !coeffs) return CSP_ERR_INVALID_ARG;
    // chunk g runs on the g-th gfx950 device (other architectures may sit between them in HIP's numbering)
    std::vector<int> ordinals;
    {
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
        for (int i = 0; i < n; ++i) {
            hipDeviceProp_t p;
            if (hipGetDeviceProperties(&p, i) == hipSuccess && std::strncmp(p.gcnArchName, "gfx950", 6) == 0) ordinals.push_back(i);
        }
    }
    const int have = (int)ordinals.size();
    if (have <= 0) return CSP_ERR_NO_DEVICE;
    if (ngpu <= 0) ngpu = have;
    if (ngpu > have) return CSP_ERR_INVALID_ARG;
    if ((int64_t)ngpu > s.B) ngpu = (int)s.B;
    if (desc->mem_space == CSP_MEM_DEVICE)   // the batch lives on a root device: scatter / solve / gather over RCCL
        return solve_sharded_device(desc, s, waypoints, times, bc, coeffs, max_dev, status, ordinals, ngpu);
    const int span_choice = use_span(desc, s) ? 1 : 0;   // from the WHOLE batch; pinned for every chunk
    const size_t m = 2 * (size_t)s.order;
    struct Chunk {
        csp_minsnap_desc d;
        std::vector<int64_t> off;   // ragged: the chunk's own prefix sums
        int rc = CSP_OK;
        std::string err;
    };
    std::vector<Chunk> ch((size_t)ngpu);
    std::vector<std::thread> th;
    for (int g = 0; g < ngpu; ++g) {
        const int64_t lo = s.B * g / ngpu, hi = s.B * (g + 1) / ngpu;   // contiguous, balanced
        const int64_t seg_lo = s.ragged ? desc->seg_offsets[lo] : lo * (int64_t)s.S;
        Chunk &c = ch[(size_t)g];
        c.d = *desc;
        c.d.batch = hi - lo;
        c.d.device_id = ordinals[(size_t)g];
        if (s.ragged) {
            c.off.resize((size_t)(hi - lo + 1));
            for (int64_t b = lo; b <= hi; ++b) c.off[(size_t)(b - lo)] = desc->seg_offsets[b] - seg_lo;
            c.d.seg_offsets = c.off.data();
        }
        if (desc->vel_zero_weight_per_traj) c.d.vel_zero_weight_per_traj = desc->vel_zero_weight_per_traj + lo;
        const char *wp = (const char *)waypoints + (size_t)(seg_lo + lo) * 3 * s.elt;
        const char *tm = (const char *)times + (size_t)seg_lo * s.elt;
        const char *bcp = (const char *)bc + (desc->bc_per_trajectory ? (size_t)lo * 12 * s.elt : 0);
        char *co = (char *)coeffs + (size_t)seg_lo * 3 * m * s.elt;
        double *md = max_dev ? max_dev + lo : nullptr;
        int32_t *stt = status ? status + lo : nullptr;
        // one host thread per device: each stages its chunk through that device's cached arena (pinned halves,
        // DMA overlapped with the CPU copy) on a stream of its own, so the devices' transfers and kernels overlap
        th.emplace_back([&c, wp, tm, bcp, co, md, stt, span_choice]() {
            g_span_override = span_choice;
            hipStream_t st = nullptr;
            if (hipSetDevice(c.d.device_id) != hipSuccess || hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) {
                c.rc = CSP_ERR_HIP;
                c.err = "hipSetDevice / hipStreamCreate failed in a shard worker";
                return;
            }
            c.rc = csp_minsnap_solve_batch(&c.d, wp, tm, bcp, co, md, stt, nullptr, 0, st);
            if (c.rc != CSP_OK) c.err = g_last_hip_error;   // the worker's thread-local text
            (void)hipStreamDestroy(st);
        });
    }
    for (auto &t : th) t.join();
    for (const Chunk &c : ch)
        if (c.rc != CSP_OK) { g_last_hip_error = c.err; return c.rc; }
    return CSP_OK;
}

int csp_minsnap_time_alloc_batch(const csp_minsnap_desc *desc, const void *waypoints, double v_avg,
                                 double min_time_s, void *times, void *hip_stream) {
    Shape s;
    int rc = validate(desc, s);
    if (rc != CSP_OK) return rc;
    if (s.B == 0) return CSP_OK;
    if (!waypoints || !times) return CSP_ERR_INVALID_ARG;
    rc = select_device(desc->device_id);
    if (rc != CSP_OK) return rc;
    hipStream_t st = (hipStream_t)hip_stream;
    csp::TimeAllocArgs a;
    a.B = s.B; a.S = s.S; a.v_avg = v_avg; a.min_time_s = min_time_s;
    if (desc->mem_space == CSP_MEM_DEVICE) {
        a.wp = waypoints; a.times = times; a.seg_off = s.ragged ? desc->seg_offsets : nullptr;
        hipError_t e = csp::launch_time_alloc(a, s.f32, st);
        return e == hipSuccess ? CSP_OK : hip_fail(e, "time_alloc launch");
    }
    const int64_t total_seg = s.ragged ? desc->seg_offsets[s.B] : s.B * (int64_t)s.S;
    const size_t n_wp = (size_t)(total_seg + s.B) * 3 * s.elt, n_tm = (size_t)total_seg * s.elt;
    csp::HostCall hc(current_device(), st);
    const size_t o_wp = hc.in(waypoints, n_wp);
    const size_t o_so = s.ragged ? hc.in(desc->seg_offsets, (size_t)(s.B + 1) * 8) : 0;
    const size_t o_tm = hc.out(times, n_tm);
    CSP_HIP(hc.upload());
    a.wp = hc.ptr(o_wp); a.times = hc.ptr(o_tm); a.seg_off = s.ragged ? hc.ptr<const int64_t>(o_so) : nullptr;
    hipError_t e = csp::launch_time_alloc(a, s.f32, st);
    if (e != hipSuccess) return hip_fail(e, "time_alloc launch");
    CSP_HIP(hc.download());
    return CSP_OK;
}

size_t csp_minsnap_plan_workspace_bytes(const csp_minsnap_desc *desc) {
    Shape s;
    if (validate(desc, s) != CSP_OK) return 0;
    csp_minsnap_desc g = *desc;
    Shape gs;
    validate(&g, gs);
    // solve workspace + vw[B] f64 + max_dev[B] f64 + iters[B] i32 + done[B] i32 + the count of unfinished trajectories (i32)
    // + t* indices [S][B] i32 (path kernel)
    const size_t tau = (desc->path_weight > 0.0 && use_fixed(&g, gs)) ? align_up((size_t)s.B * (size_t)s.S * 4, 256) : 0;
    return align_up(ws_bytes(&g, gs, nullptr), 256) + align_up((size_t)s.B * 8, 256) * 2 + align_up((size_t)s.B * 4, 256) * 2 + 256 + tau;
}

}  // extern "C"

namespace {

// Device-memory form of csp_minsnap_plan_batch.  `sync_early_exit`: the CSP_MEM_HOST wrapper is synchronous anyway, so it
// reads the count of unfinished trajectories after every pass and stops the <= 11-pass loop as soon as it is zero (one
// flight usually converges at its first solve: ten solves saved); device-memory callers get the fully asynchronous form.
int plan_device(const csp_minsnap_desc *desc, const Shape &s, const void *waypoints, double v_avg, double min_time_s,
                const void *bc, void *times, void *coeffs, double *max_dev, double *vel_zero_weight_out,
                int32_t *iterations, int32_t *status, void *workspace, size_t workspace_bytes, hipStream_t st,
                bool sync_early_exit, int phase = 0, int32_t *pending_ext = nullptr, int32_t **done_out = nullptr) {
    // phase 0: everything (device-memory callers; host callers that check after every pass).  The host wrapper splits the
    // call so that the common case -- every trajectory converged at its FIRST solve -- costs one synchronisation:
    // phase 1 = time allocation + first solve + bookkeeping, nothing synchronised (the "increases so far" counter sits in
    // the caller's output block and comes back with the results); phase 2 = the remaining <= 10 passes, checked per pass.
    int rc;
    hipError_t e;
    csp::TimeAllocArgs ta;
    ta.wp = waypoints; ta.times = times; ta.seg_off = s.ragged ? desc->seg_offsets : nullptr;
    ta.B = s.B; ta.S = s.S; ta.v_avg = v_avg; ta.min_time_s = min_time_s;
    if (phase != 2 && !(desc->path_weight > 0.0)) {   // with the loop: fused with the loop's initial state, below
        e = csp::launch_time_alloc(ta, s.f32, st);
        if (e != hipSuccess) return hip_fail(e, "time_alloc launch");
    }

    if (!(desc->path_weight > 0.0)) {
        // without the path penalty the deviation metric is identically 0 (t* = 0, :342), so the
        // loop of :80-90 ends after its first solve
        rc = dispatch(desc, s, waypoints, times, bc, coeffs, max_dev, status, desc->seg_offsets,
                      desc->vel_zero_weight_per_traj, workspace, workspace_bytes, st);
        if (rc != CSP_OK) return rc;
        if (iterations) CSP_HIP(hipMemsetAsync(iterations, 0, (size_t)s.B * 4, st));
        if (vel_zero_weight_out) {
            if (desc->vel_zero_weight_per_traj)
                CSP_HIP(hipMemcpyAsync(vel_zero_weight_out, desc->vel_zero_weight_per_traj, (size_t)s.B * 8, hipMemcpyDeviceToDevice, st));
            else if ((e = csp::launch_fill_f64(vel_zero_weight_out, desc->vel_zero_weight, s.B, st)) != hipSuccess)
                return hip_fail(e, "fill");
        }
        return CSP_OK;
    }

    csp_minsnap_desc g = *desc;  // the re-solve loop drives the per-trajectory weight array
    Shape gs;
    validate(&g, gs);
    const size_t need = csp_minsnap_plan_workspace_bytes(desc);
    if (!workspace || workspace_bytes < need) return CSP_ERR_WORKSPACE;
    const size_t solve_ws = align_up(ws_bytes(&g, gs, nullptr), 256);
    char *base = (char *)workspace + solve_ws;
    // the loop's per-trajectory state lives in the caller's output arrays where they were passed (no copy at the end)
    double *vw = vel_zero_weight_out ? vel_zero_weight_out : (double *)base;   base += align_up((size_t)s.B * 8, 256);
    double *md = max_dev ? max_dev : (double *)base;                           base += align_up((size_t)s.B * 8, 256);
    int32_t *iters = iterations ? iterations : (int32_t *)base;                base += align_up((size_t)s.B * 4, 256);
    int32_t *done = (int32_t *)base;                              base += align_up((size_t)s.B * 4, 256);
    int32_t *pending = pending_ext ? pending_ext : (int32_t *)base;   base += 256;   // cumulative count of weight increases
    // the pre-solve does not depend on vel_zero_weight: the first pass stores its t* indices, the others reuse them
    int *tau_buf = use_fixed(&g, gs) ? (int *)base : nullptr;
    if (phase != 2) {
        // per-trajectory starting weights: the init kernel leaves vw alone (vel_zero_weight_out may BE the caller's weight
        // array -- an in-place update -- and must not be overwritten with the scalar first), then they are copied in
        // unless they are already there
        const bool per = desc->vel_zero_weight_per_traj != nullptr;
        if ((e = csp::launch_time_alloc_init(ta, s.f32, per ? nullptr : vw, iters, done, pending, desc->vel_zero_weight, st)) != hipSuccess)
            return hip_fail(e, "time_alloc launch");
        if (per && vw != desc->vel_zero_weight_per_traj)
            CSP_HIP(hipMemcpyAsync(vw, desc->vel_zero_weight_per_traj, (size_t)s.B * 8, hipMemcpyDeviceToDevice, st));
    }
    const bool count = sync_early_exit || phase != 0;
    int32_t increased_before = 0;
    if (phase == 2) {   // the counter after the first pass (the host wrapper has it too; read here to stay self-contained)
        CSP_HIP(hipMemcpyAsync(&increased_before, pending, 4, hipMemcpyDeviceToHost, st));
        CSP_HIP(hipStreamSynchronize(st));
    }
    const int first = phase == 2 ? 1 : 0, last = phase == 1 ? 0 : 10;
    for (int pass = first; pass <= last; ++pass) {  // at most 11 solves (:78-90)
        rc = dispatch(&g, gs, waypoints, times, bc, coeffs, md, status, desc->seg_offsets, vw, workspace, solve_ws, st, done,
                      tau_buf, pass == 0 ? 1 : 2);
        if (rc != CSP_OK) return rc;
        // `done_out` (phase 1 only): the caller's next kernel does this pass's update (sample placement, one launch less)
        if (phase == 1 && done_out) *done_out = done;
        else if ((e = csp::launch_resolve_update(md, vw, iters, done, count ? pending : nullptr, s.B, st)) != hipSuccess)
            return hip_fail(e, "resolve_update");
        if (sync_early_exit && phase != 1) {
            int32_t increased = 0;   // cumulative: a pass that raised nobody's weight was the last one anybody needed
            CSP_HIP(hipMemcpyAsync(&increased, pending, 4, hipMemcpyDeviceToHost, st));
            CSP_HIP(hipStreamSynchronize(st));
            if (increased == increased_before) break;
            increased_before = increased;
        }
    }
    return CSP_OK;
}

}  // namespace

extern "C" {

int csp_minsnap_plan_batch(const csp_minsnap_desc *desc, const void *waypoints, double v_avg, double min_time_s,
                           const void *bc, void *times, void *coeffs, double *max_dev, double *vel_zero_weight_out,
                           int32_t *iterations, int32_t *status, void *workspace, size_t workspace_bytes,
                           void *hip_stream) {
    Shape s;
    int rc = validate(desc, s);
    if (rc != CSP_OK) return rc;
    if (s.B == 0) return CSP_OK;
    if (!waypoints || !bc || !times || !coeffs) return CSP_ERR_INVALID_ARG;
    rc = select_device(desc->device_id);
    if (rc != CSP_OK) return rc;
    hipStream_t st = (hipStream_t)hip_stream;
    if (desc->mem_space == CSP_MEM_DEVICE)
        return plan_device(desc, s, waypoints, v_avg, min_time_s, bc, times, coeffs, max_dev, vel_zero_weight_out, iterations,
                           status, workspace, workspace_bytes, st, false);

    // CSP_MEM_HOST: stage through the device's cached arena, then run the device-memory form
    const int64_t total_seg = s.ragged ? desc->seg_offsets[s.B] : s.B * (int64_t)s.S;
    const size_t m = 2 * (size_t)s.order;
    const size_t n_wp = (size_t)(total_seg + s.B) * 3 * s.elt, n_tm = (size_t)total_seg * s.elt;
    const size_t n_bc = (size_t)(desc->bc_per_trajectory ? s.B : 1) * 12 * s.elt, n_co = (size_t)total_seg * 3 * m * s.elt;
    csp_minsnap_desc dd = *desc;
    dd.mem_space = CSP_MEM_DEVICE;
    const size_t n_ws = csp_minsnap_plan_workspace_bytes(&dd);
    csp::HostCall hc(current_device(), st);
    const size_t o_wp = hc.in(waypoints, n_wp), o_bc = hc.in(bc, n_bc);
    const size_t o_so = s.ragged ? hc.in(desc->seg_offsets, (size_t)(s.B + 1) * 8) : 0;
    const size_t o_vi = desc->vel_zero_weight_per_traj ? hc.in(desc->vel_zero_weight_per_traj, (size_t)s.B * 8) : 0;
    const size_t o_tm = hc.out(times, n_tm), o_co = hc.out(coeffs, n_co);
    const size_t o_md = hc.out(max_dev, (size_t)s.B * 8), o_vw = hc.out(vel_zero_weight_out, (size_t)s.B * 8);
    const size_t o_it = hc.out(iterations, (size_t)s.B * 4), o_st = hc.out(status, (size_t)s.B * 4);
    int32_t raised = 0;
    const size_t o_pd = hc.out(&raised, 4);
    const size_t o_ws = hc.scratch(n_ws);
    CSP_HIP(hc.upload());
    if (s.ragged) dd.seg_offsets = hc.ptr<const int64_t>(o_so);
    dd.vel_zero_weight_per_traj = desc->vel_zero_weight_per_traj ? hc.ptr<const double>(o_vi) : nullptr;
    if (!(desc->path_weight > 0.0)) {   // one solve, no loop
        rc = plan_device(&dd, s, hc.ptr(o_wp), v_avg, min_time_s, hc.ptr(o_bc), hc.ptr(o_tm), hc.ptr(o_co), hc.ptr<double>(o_md),
                         hc.ptr<double>(o_vw), hc.ptr<int32_t>(o_it), hc.ptr<int32_t>(o_st), hc.ptr(o_ws), n_ws, st, true);
        if (rc != CSP_OK) return rc;
        CSP_HIP(hc.download());
        return CSP_OK;
    }
    // first solve, results and the "weights raised so far" counter in ONE copy back; only if somebody's weight was raised do
    // the remaining passes run (checked one by one) and the results come back a second time
    rc = plan_device(&dd, s, hc.ptr(o_wp), v_avg, min_time_s, hc.ptr(o_bc), hc.ptr(o_tm), hc.ptr(o_co), hc.ptr<double>(o_md),
                     hc.ptr<double>(o_vw), hc.ptr<int32_t>(o_it), hc.ptr<int32_t>(o_st), hc.ptr(o_ws), n_ws, st, true, 1,
                     hc.ptr<int32_t>(o_pd));
    if (rc != CSP_OK) return rc;
    CSP_HIP(hc.download());
    if (raised != 0) {
        hc.touch();
        rc = plan_device(&dd, s, hc.ptr(o_wp), v_avg, min_time_s, hc.ptr(o_bc), hc.ptr(o_tm), hc.ptr(o_co), hc.ptr<double>(o_md),
                         hc.ptr<double>(o_vw), hc.ptr<int32_t>(o_it), hc.ptr<int32_t>(o_st), hc.ptr(o_ws), n_ws, st, true, 2,
                         hc.ptr<int32_t>(o_pd));
        if (rc != CSP_OK) return rc;
        CSP_HIP(hc.download());
    }
    return CSP_OK;
}

int csp_minsnap_sample_batch(const csp_minsnap_desc *desc, const void *times, const void *coeffs,
                             double sample_distance, int64_t capacity, void *samples, int32_t *counts,
                             double *stats, void *hip_stream) {
    Shape s;
    int rc = validate(desc, s);
    if (rc != CSP_OK) return rc;
    if (s.B == 0) return CSP_OK;
    if (!times || !coeffs || !samples || !counts || capacity < 1) return CSP_ERR_INVALID_ARG;
    rc = select_device(desc->device_id);
    if (rc != CSP_OK) return rc;
    hipStream_t st = (hipStream_t)hip_stream;
    csp::SampleArgs a;
    a.B = s.B; a.S = s.S; a.order = s.order; a.capacity = capacity; a.sample_distance = sample_distance;
    a.keep_dist2 = keep_threshold(sample_distance);
    a.seg_major = (desc->flags & CSP_FLAG_SEGMENT_MAJOR) ? 1 : 0;
    a.Smax = s.Smax;
    a.one_lane = (desc->flags & CSP_FLAG_FORCE_GENERIC) ? 1 : 0;
    a.long_segments = (desc->flags & CSP_FLAG_LONG_SEGMENTS) ? 1 : 0;
    if (desc->mem_space == CSP_MEM_DEVICE) {
        a.times = times; a.coeffs = coeffs; a.seg_off = s.ragged ? desc->seg_offsets : nullptr;
        a.samples = samples; a.counts = counts; a.stats = stats;
        hipError_t e = csp::launch_sample(a, s.f32, st);
        return e == hipSuccess ? CSP_OK : hip_fail(e, "sample launch");
    }
    const int64_t total_seg = s.ragged ? desc->seg_offsets[s.B] : s.B * (int64_t)s.S;
    if (!a.long_segments && total_seg > 0) {
        // host-resident times: count the candidates (dt = min(0.1, T/10), :126) and hand long legs --
        // more than 128 evaluations per segment on average -- to the wave-cooperative sampler
        double cand = 0.0;
        for (int64_t i = 0; i < total_seg; ++i) {
            const double T = s.f32 ? (double)((const float *)times)[i] : ((const double *)times)[i];
            cand += (T > 1.0 && T < 1e12) ? T * 10.0 : 10.0;
        }
        a.long_segments = cand > 128.0 * (double)total_seg;
    }
    const size_t m = 2 * (size_t)s.order;
    const size_t n_tm = (size_t)total_seg * s.elt, n_co = (size_t)total_seg * 3 * m * s.elt;
    const size_t n_sm = (size_t)s.B * (size_t)capacity * 3 * s.elt;
    // A few long flights (the reference's own call: ONE flight of kilometre legs): one wave per SEGMENT into per-segment
    // runs, then placement + end-point rule + statistics (minsnap_plan.hip sample_wave_seg_kernel).  Needs fp64 storage,
    // the default layout and a `capacity` that holds every candidate (the class shim passes exactly that), so that no run
    // and no trajectory can overflow.
    std::vector<int64_t> run_off;
    if (a.long_segments && !a.one_lane && !s.f32 && !a.seg_major && s.B <= 64 && total_seg >= 2) {
        run_off.resize((size_t)total_seg + 1);
        run_off[0] = 0;
        bool fits = true;
        int64_t g = 0;
        for (int64_t b = 0; b < s.B && fits; ++b) {
            const int64_t nseg = s.ragged ? desc->seg_offsets[b + 1] - desc->seg_offsets[b] : s.S;
            int64_t traj = 2;
            for (int64_t k = 0; k < nseg; ++k, ++g) {
                const double T = ((const double *)times)[g];
                int64_t cand = 0;
                if (T >= 1.0e-14 && T <= 1.0e7) cand = (int64_t)((T + 1e-12) / (T / 10.0 < 0.1 ? T / 10.0 : 0.1)) + 2;
                run_off[(size_t)g + 1] = run_off[(size_t)g] + cand + 3;   // + first sample, + the parked end point, + slack
                traj += cand;
            }
            fits = traj <= capacity;
        }
        if (!fits) run_off.clear();
    }
    csp::HostCall hc(current_device(), st);
    const size_t o_tm = hc.in(times, n_tm), o_co = hc.in(coeffs, n_co);
    const size_t o_so = s.ragged ? hc.in(desc->seg_offsets, (size_t)(s.B + 1) * 8) : 0;
    const size_t o_ro = !run_off.empty() ? hc.in(run_off.data(), run_off.size() * 8) : 0;
    const size_t o_ct = hc.out(counts, (size_t)s.B * 4), o_sx = hc.out(stats, (size_t)s.B * 16);
    // one flight: `capacity` is an upper bound (every candidate).  Up to 1 MB it simply comes back with the count and the
    // statistics in ONE copy (a second round trip costs more than the surplus rows); beyond that the count is fetched first
    // and only the rows in use follow
    const bool two_step = s.B == 1 && n_sm > ((size_t)1 << 20);
    const size_t o_sm = hc.out(two_step ? nullptr : samples, n_sm);
    const size_t o_tmp = !run_off.empty() ? hc.scratch((size_t)run_off.back() * 24) : 0;
    const size_t o_sc = !run_off.empty() ? hc.scratch((size_t)total_seg * 4) : 0;
    CSP_HIP(hc.upload());
    a.times = hc.ptr(o_tm); a.coeffs = hc.ptr(o_co); a.seg_off = s.ragged ? hc.ptr<const int64_t>(o_so) : nullptr;
    a.samples = hc.ptr(o_sm); a.counts = hc.ptr<int32_t>(o_ct); a.stats = hc.ptr<double>(o_sx);
    hipError_t e;
    if (!run_off.empty()) {
        e = csp::launch_sample_segment_waves(a, hc.ptr<double>(o_tmp), hc.ptr<const int64_t>(o_ro), hc.ptr<int32_t>(o_sc), total_seg, st);
    } else {
        e = csp::launch_sample(a, s.f32, st);
    }
    if (e != hipSuccess) return hip_fail(e, "sample launch");
    if (two_step) {
        CSP_HIP(hipMemcpyAsync(counts, hc.ptr(o_ct), 4, hipMemcpyDeviceToHost, st));
        CSP_HIP(hipStreamSynchronize(st));
        const int64_t rows = counts[0] < capacity ? (counts[0] > 0 ? counts[0] : 0) : capacity;
        if (rows) CSP_HIP(hipMemcpyAsync(samples, hc.ptr(o_sm), (size_t)rows * 3 * s.elt, hipMemcpyDeviceToHost, st));
        if (stats) CSP_HIP(hipMemcpyAsync(stats, hc.ptr(o_sx), 16, hipMemcpyDeviceToHost, st));
        CSP_HIP(hipStreamSynchronize(st));
        return CSP_OK;
    }
    CSP_HIP(hc.download());
    return CSP_OK;
}

}  // extern "C"

namespace {

// Host-side estimate of the segment times (minimum_snap.cpp:59-72) for SIZING only: the device computes them again
// (possibly an ulp away: fused multiply-adds), so every count derived from these carries slack.
void estimate_times(const csp_minsnap_desc *desc, const Shape &s, const void *waypoints, double v_avg, double min_time_s,
                    std::vector<double> &T) {
    const int64_t total_seg = s.ragged ? desc->seg_offsets[s.B] : s.B * (int64_t)s.S;
    T.resize((size_t)total_seg);
    int64_t g = 0;
    for (int64_t b = 0; b < s.B; ++b) {
        const int64_t nseg = s.ragged ? desc->seg_offsets[b + 1] - desc->seg_offsets[b] : s.S;
        for (int64_t k = 0; k < nseg; ++k, ++g) {
            double p[6];
            for (int q = 0; q < 6; ++q)
                p[q] = s.f32 ? (double)((const float *)waypoints)[(g + b) * 3 + q] : ((const double *)waypoints)[(g + b) * 3 + q];
            const double dx = p[3] - p[0], dy = p[4] - p[1], dz = p[5] - p[2];
            double t = v_avg > 1e-6 ? std::sqrt(dx * dx + dy * dy + dz * dz) / v_avg : min_time_s;
            if (t < min_time_s) t = min_time_s;
            T[(size_t)g] = t;
        }
    }
}

// Candidates of one segment (:126-136), generously: the device's time may differ in its last bits
int64_t candidate_bound(double T, bool f32) {
    if (!(T >= 1.0e-15 && T <= 1.1e7)) return 0;   // the device emits none outside [1e-14, 1e7] (minsnap_plan.hip t_end)
    const double Tw = T * (1.0 + (f32 ? 1e-6 : 1e-12));
    const double dt = T / 10.0 < 0.1 ? T / 10.0 : 0.1;
    return (int64_t)((Tw + 1e-12) / (dt * (1.0 - (f32 ? 1e-6 : 1e-12)))) + 2;
}

}  // namespace

extern "C" {

int64_t csp_minsnap_sample_capacity(const csp_minsnap_desc *desc, const void *waypoints_host, double v_avg, double min_time_s) {
    Shape s;
    if (validate(desc, s) != CSP_OK || !waypoints_host) return -1;
    std::vector<double> T;
    estimate_times(desc, s, waypoints_host, v_avg, min_time_s, T);
    int64_t cap = 2, g = 0;
    for (int64_t b = 0; b < s.B; ++b) {
        const int64_t nseg = s.ragged ? desc->seg_offsets[b + 1] - desc->seg_offsets[b] : s.S;
        int64_t traj = 2;
        for (int64_t k = 0; k < nseg; ++k, ++g) traj += candidate_bound(T[(size_t)g], s.f32);
        if (traj > cap) cap = traj;
    }
    return cap;
}

int csp_minsnap_generate_batch(const csp_minsnap_desc *desc, const void *waypoints, double v_avg, double min_time_s,
                               const void *bc, double sample_distance, int64_t capacity, void *samples,
                               int32_t *counts, double *stats, void *times, void *coeffs, double *max_dev,
                               double *vel_zero_weight_out, int32_t *iterations, int32_t *status,
                               void *workspace, size_t workspace_bytes, void *hip_stream) {
    Shape s;
    int rc = validate(desc, s);
    if (rc != CSP_OK) return rc;
    if (s.B == 0) return CSP_OK;
    if (!waypoints || !bc || !samples || !counts || capacity < 1) return CSP_ERR_INVALID_ARG;
    if (desc->flags & CSP_FLAG_SEGMENT_MAJOR) return CSP_ERR_INVALID_ARG;   // the plan call has no segment-major form
    rc = select_device(desc->device_id);
    if (rc != CSP_OK) return rc;
    hipStream_t st = (hipStream_t)hip_stream;
    csp::SampleArgs a;
    a.B = s.B; a.S = s.S; a.order = s.order; a.capacity = capacity; a.sample_distance = sample_distance;
    a.keep_dist2 = keep_threshold(sample_distance);
    a.seg_major = 0;
    a.Smax = s.Smax;
    a.one_lane = (desc->flags & CSP_FLAG_FORCE_GENERIC) ? 1 : 0;
    a.long_segments = (desc->flags & CSP_FLAG_LONG_SEGMENTS) ? 1 : 0;
    if (desc->mem_space == CSP_MEM_DEVICE) {
        if (!times || !coeffs) return CSP_ERR_INVALID_ARG;
        rc = plan_device(desc, s, waypoints, v_avg, min_time_s, bc, times, coeffs, max_dev, vel_zero_weight_out, iterations,
                         status, workspace, workspace_bytes, st, false);
        if (rc != CSP_OK) return rc;
        a.times = times; a.coeffs = coeffs; a.seg_off = s.ragged ? desc->seg_offsets : nullptr;
        a.samples = samples; a.counts = counts; a.stats = stats;
        hipError_t e = csp::launch_sample(a, s.f32, st);
        return e == hipSuccess ? CSP_OK : hip_fail(e, "sample launch");
    }

    // CSP_MEM_HOST: one arena, one upload, [time allocation, first solve, loop bookkeeping, sampling], one download
    const int64_t total_seg = s.ragged ? desc->seg_offsets[s.B] : s.B * (int64_t)s.S;
    std::vector<double> Test;
    estimate_times(desc, s, waypoints, v_avg, min_time_s, Test);
    if (!a.long_segments && total_seg > 0) {   // as csp_minsnap_sample_batch decides from host-resident times
        double cand = 0.0;
        for (double T : Test) cand += (T > 1.0 && T < 1e12) ? T * 10.0 : 10.0;
        a.long_segments = cand > 128.0 * (double)total_seg;
    }
    std::vector<int64_t> run_off;   // per-segment runs of the one-wave-per-segment sampler (see csp_minsnap_sample_batch)
    if (a.long_segments && !a.one_lane && !s.f32 && s.B <= 64 && total_seg >= 2) {
        run_off.resize((size_t)total_seg + 1);
        run_off[0] = 0;
        bool fits = true;
        int64_t g = 0;
        for (int64_t b = 0; b < s.B && fits; ++b) {
            const int64_t nseg = s.ragged ? desc->seg_offsets[b + 1] - desc->seg_offsets[b] : s.S;
            int64_t traj = 2;
            for (int64_t k = 0; k < nseg; ++k, ++g) {
                const int64_t cand = candidate_bound(Test[(size_t)g], false);
                run_off[(size_t)g + 1] = run_off[(size_t)g] + cand + 3;   // + first sample, + the parked end point, + slack
                traj += cand;
            }
            fits = traj <= capacity;
        }
        if (!fits) run_off.clear();
    }
    const size_t m = 2 * (size_t)s.order;
    const size_t n_wp = (size_t)(total_seg + s.B) * 3 * s.elt, n_tm = (size_t)total_seg * s.elt;
    const size_t n_bc = (size_t)(desc->bc_per_trajectory ? s.B : 1) * 12 * s.elt, n_co = (size_t)total_seg * 3 * m * s.elt;
    const size_t n_sm = (size_t)s.B * (size_t)capacity * 3 * s.elt;
    csp_minsnap_desc dd = *desc;
    dd.mem_space = CSP_MEM_DEVICE;
    const size_t n_ws = csp_minsnap_plan_workspace_bytes(&dd);
    // one flight with a large `capacity` (an upper bound: every candidate): beyond 1 MB only the rows in use come back
    const bool two_step = s.B == 1 && n_sm > ((size_t)1 << 20);
    csp::HostCall hc(current_device(), st);
    const size_t o_wp = hc.in(waypoints, n_wp), o_bc = hc.in(bc, n_bc);
    const size_t o_so = s.ragged ? hc.in(desc->seg_offsets, (size_t)(s.B + 1) * 8) : 0;
    const size_t o_vi = desc->vel_zero_weight_per_traj ? hc.in(desc->vel_zero_weight_per_traj, (size_t)s.B * 8) : 0;
    const size_t o_ro = !run_off.empty() ? hc.in(run_off.data(), run_off.size() * 8) : 0;
    int32_t raised = 0;
    const size_t o_ct = hc.out(counts, (size_t)s.B * 4), o_sx = hc.out(stats, (size_t)s.B * 16), o_pd = hc.out(&raised, 4);
    const size_t o_md = hc.out(max_dev, (size_t)s.B * 8), o_vw = hc.out(vel_zero_weight_out, (size_t)s.B * 8);
    const size_t o_it = hc.out(iterations, (size_t)s.B * 4), o_st = hc.out(status, (size_t)s.B * 4);
    const size_t o_sm_out = two_step ? 0 : hc.out(samples, n_sm);
    const size_t o_tm_out = times ? hc.out(times, n_tm) : 0, o_co_out = coeffs ? hc.out(coeffs, n_co) : 0;
    const size_t o_sm = two_step ? hc.scratch(n_sm) : o_sm_out;
    const size_t o_tm = times ? o_tm_out : hc.scratch(n_tm), o_co = coeffs ? o_co_out : hc.scratch(n_co);
    const size_t o_ws = hc.scratch(n_ws);
    const size_t o_tmp = !run_off.empty() ? hc.scratch((size_t)run_off.back() * 24) : 0;
    const size_t o_sc = !run_off.empty() ? hc.scratch((size_t)total_seg * 4) : 0;
    CSP_HIP(hc.upload());
    if (s.ragged) dd.seg_offsets = hc.ptr<const int64_t>(o_so);
    dd.vel_zero_weight_per_traj = desc->vel_zero_weight_per_traj ? hc.ptr<const double>(o_vi) : nullptr;
    a.times = hc.ptr(o_tm); a.coeffs = hc.ptr(o_co); a.seg_off = s.ragged ? hc.ptr<const int64_t>(o_so) : nullptr;
    a.samples = hc.ptr(o_sm); a.counts = hc.ptr<int32_t>(o_ct); a.stats = hc.ptr<double>(o_sx);
    const bool loop = desc->path_weight > 0.0;
    for (int round = 0; round < 2; ++round) {
        // round 0: time allocation + first solve (+ bookkeeping); round 1 (only if somebody's weight was raised): the
        // remaining <= 10 passes, checked one by one.  Sampling and ONE copy back follow either.
        if (round) hc.touch();
        int32_t *done_dev = nullptr;   // first pass of the loop + per-segment sampler: the placement kernel does the pass's update
        const bool fold = loop && round == 0 && !run_off.empty();
        rc = plan_device(&dd, s, hc.ptr(o_wp), v_avg, min_time_s, hc.ptr(o_bc), hc.ptr(o_tm), hc.ptr(o_co), hc.ptr<double>(o_md),
                         hc.ptr<double>(o_vw), hc.ptr<int32_t>(o_it), hc.ptr<int32_t>(o_st), hc.ptr(o_ws), n_ws, st, true,
                         loop ? round + 1 : 0, loop ? hc.ptr<int32_t>(o_pd) : nullptr, fold ? &done_dev : nullptr);
        if (rc != CSP_OK) return rc;
        hipError_t e;
        if (!run_off.empty()) {
            const csp::LoopUpdate upd = {hc.ptr<double>(o_md), hc.ptr<double>(o_vw), hc.ptr<int32_t>(o_it), done_dev, hc.ptr<int32_t>(o_pd)};
            e = csp::launch_sample_segment_waves_upd(a, hc.ptr<double>(o_tmp), hc.ptr<const int64_t>(o_ro), hc.ptr<int32_t>(o_sc), total_seg,
                                                     fold && done_dev ? &upd : nullptr, st);
        } else
            e = csp::launch_sample(a, s.f32, st);
        if (e != hipSuccess) return hip_fail(e, "sample launch");
        CSP_HIP(hc.download());
        if (!loop || raised == 0) break;
    }
    if (two_step) {
        const int64_t rows = counts[0] < capacity ? (counts[0] > 0 ? counts[0] : 0) : capacity;
        if (rows) {
            CSP_HIP(hipMemcpyAsync(samples, hc.ptr(o_sm), (size_t)rows * 3 * s.elt, hipMemcpyDeviceToHost, st));
            CSP_HIP(hipStreamSynchronize(st));
        }
    }
    return CSP_OK;
}

void csp_minsnap_release_cached_memory(void) { csp::arena_free_idle(); }

}  // extern "C"
