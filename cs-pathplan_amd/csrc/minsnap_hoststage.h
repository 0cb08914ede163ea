// minsnap_hoststage.h -- staging for CSP_MEM_HOST callers of the C-ABI (the reference's own call pattern is
// ONE flight per call from host memory: uavPathPlanning.cpp:4423, :4461).
//
// A host-memory call needs device images of its inputs and outputs.  Allocating them per call costs more
// than the kernels (hipMalloc x 9 + hipFree x 9, the latter synchronising the device, + pageable copies),
// so every device keeps a small pool of ARENAS: one grow-only device allocation, 16 MB of page-locked
// staging memory and two events.  A call borrows an arena for its duration (exclusive), carves all of its
// buffers out of the one allocation (inputs first, then outputs, then scratch -- so that small calls move
// their inputs with ONE host-to-device copy and their outputs with ONE device-to-host copy through the
// pinned block) and hands it back; nothing is freed until csp_minsnap_release_cached_memory().
// Large transfers stream through the pinned block in two 8 MB halves: the DMA of one half overlaps the
// CPU copy of the other (itself spread over a few helper threads, CopyPool); caller buffers that are
// page-locked already are handed to the DMA engine directly.
#pragma once
#include <hip/hip_runtime.h>

#include <condition_variable>
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace csp {

struct Arena {
    static constexpr size_t PIN_BYTES = 16u << 20, HALF = 8u << 20;
    int device = -1;
    char *dev = nullptr;
    size_t dev_cap = 0;
    char *pin = nullptr;
    hipEvent_t ev[2] = {nullptr, nullptr};

    hipError_t reserve(size_t n) {
        if (n <= dev_cap) return hipSuccess;
        size_t want = n + n / 4;                       // grow by at least a quarter: a planner's batches creep
        want = (want + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
        if (dev) { (void)hipFree(dev); dev = nullptr; dev_cap = 0; }
        hipError_t e = hipMalloc((void **)&dev, want);
        if (e != hipSuccess) { dev = nullptr; return e; }
        dev_cap = want;
        return hipSuccess;
    }
    hipError_t init_pin() {
        if (pin) return hipSuccess;
        hipError_t e = hipHostMalloc((void **)&pin, PIN_BYTES, hipHostMallocDefault);
        if (e != hipSuccess) { pin = nullptr; return e; }
        for (int i = 0; i < 2; ++i)
            if ((e = hipEventCreateWithFlags(&ev[i], hipEventDisableTiming)) != hipSuccess) return e;
        return hipSuccess;
    }
    void destroy() {
        if (dev) (void)hipFree(dev);
        if (pin) (void)hipHostFree(pin);
        for (int i = 0; i < 2; ++i)
            if (ev[i]) (void)hipEventDestroy(ev[i]);
        dev = pin = nullptr;
        dev_cap = 0;
        ev[0] = ev[1] = nullptr;
    }
};

struct ArenaPool {
    std::mutex m;
    std::vector<Arena *> idle;     // arenas are leaked at process exit on purpose: the HIP runtime may already be gone
    std::vector<Arena *> all;
};

inline ArenaPool &arena_pool(int device) {
    static ArenaPool pools[64];
    return pools[device >= 0 && device < 64 ? device : 63];
}

inline Arena *arena_acquire(int device) {
    ArenaPool &p = arena_pool(device);
    std::lock_guard<std::mutex> g(p.m);
    if (!p.idle.empty()) {
        Arena *a = p.idle.back();
        p.idle.pop_back();
        return a;
    }
    Arena *a = new Arena();
    a->device = device;
    p.all.push_back(a);
    return a;
}

// What an idle arena may keep of device memory (a call that needed more gives it back on release): one 1.9 GB host batch
// would otherwise leave 2.4 GB of HBM cached per concurrently used arena for the life of a planner process.
// CSP_ARENA_KEEP_MB overrides (default 512).
inline size_t arena_keep_bytes() {
    static const size_t keep = [] {
        const char *e = std::getenv("CSP_ARENA_KEEP_MB");
        const long mb = e ? std::atol(e) : 512;
        return (size_t)(mb > 0 ? mb : 0) << 20;
    }();
    return keep;
}

inline void arena_release(Arena *a) {
    if (a->dev_cap > arena_keep_bytes() && a->dev) {   // the releasing thread's current device is the arena's (HostCall, RcclTransport)
        (void)hipFree(a->dev);
        a->dev = nullptr;
        a->dev_cap = 0;
    }
    ArenaPool &p = arena_pool(a->device);
    std::lock_guard<std::mutex> g(p.m);
    p.idle.push_back(a);
}

// Frees every idle arena of every device (arenas borrowed by calls in flight are left alone).
inline void arena_free_idle() {
    int cur = 0;
    const bool have_cur = hipGetDevice(&cur) == hipSuccess;
    for (int d = 0; d < 64; ++d) {
        ArenaPool &p = arena_pool(d);
        std::lock_guard<std::mutex> g(p.m);
        if (p.idle.empty()) continue;
        if (hipSetDevice(d) != hipSuccess) continue;
        for (Arena *a : p.idle) {
            a->destroy();
            for (size_t i = 0; i < p.all.size(); ++i)
                if (p.all[i] == a) { p.all.erase(p.all.begin() + (long)i); break; }
            delete a;
        }
        p.idle.clear();
    }
    if (have_cur) (void)hipSetDevice(cur);
}

// Staging copies of large batches are CPU-bound on one core (~10 GB/s against ~50 GB/s of PCIe): a few helper threads
// (started at the first large transfer, idle on a condition variable otherwise, never joined -- the process may be
// exiting when the library is unloaded) copy an 8 MB half in parallel slices.
class CopyPool {
public:
    static CopyPool &get() { static CopyPool *p = new CopyPool(); return *p; }
    // memcpy(dst, src, n) over the calling thread and the helpers; callers are serialised (one arena streams at a time)
    void copy(void *dst, const void *src, size_t n) {
        const size_t parts = workers_ + 1, piece = ((n / parts) + 4095) & ~(size_t)4095;
        if (workers_ == 0 || n < ((size_t)1 << 20) || piece == 0) { std::memcpy(dst, src, n); return; }
        std::lock_guard<std::mutex> serial(call_);
        {
            std::lock_guard<std::mutex> g(m_);
            dst_ = (char *)dst; src_ = (const char *)src; n_ = n; piece_ = piece;
            pending_ = (int)workers_;
            ++generation_;
        }
        cv_.notify_all();
        slice(0);
        std::unique_lock<std::mutex> g(m_);
        done_.wait(g, [&] { return pending_ == 0; });
    }

private:
    CopyPool() {
        const unsigned hw = std::thread::hardware_concurrency();
        workers_ = hw >= 8 ? 3 : (hw >= 4 ? 1 : 0);
        for (size_t w = 0; w < workers_; ++w) std::thread([this, w] { run(w + 1); }).detach();
    }
    void slice(size_t k) {
        const size_t lo = k * piece_;
        if (lo >= n_) return;
        const size_t len = (k == workers_) ? n_ - lo : (lo + piece_ <= n_ ? piece_ : n_ - lo);
        std::memcpy(dst_ + lo, src_ + lo, len);
    }
    void run(size_t k) {
        unsigned long seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> g(m_);
                cv_.wait(g, [&] { return generation_ != seen; });
                seen = generation_;
            }
            slice(k);
            {
                std::lock_guard<std::mutex> g(m_);
                if (--pending_ == 0) done_.notify_one();
            }
        }
    }
    std::mutex m_, call_;
    std::condition_variable cv_, done_;
    size_t workers_ = 0, n_ = 0, piece_ = 0;
    char *dst_ = nullptr;
    const char *src_ = nullptr;
    int pending_ = 0;
    unsigned long generation_ = 0;
};

// Page-locked (hipHostMalloc / hipHostRegister) caller memory needs no staging: the DMA engine reads and writes it directly.
inline bool host_pinned(const void *p) {
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return at.type == hipMemoryTypeHost;
}

// One host-memory call: register the buffers, upload(), launch on ptr(offset), download().
class HostCall {
public:
    HostCall(int device, hipStream_t st) : a_(arena_acquire(device)), st_(st) {}
    ~HostCall() {
        if (dirty_) (void)hipStreamSynchronize(st_);   // an error path left copies in flight: the arena must be quiet
        arena_release(a_);
    }
    HostCall(const HostCall &) = delete;
    HostCall &operator=(const HostCall &) = delete;

    // All in() calls first, then out(), then scratch(): the three groups are contiguous in the arena.
    size_t in(const void *host, size_t bytes) { ins_.push_back({(void *)host, top_, bytes}); return bump(bytes, in_end_); }
    size_t out(void *host, size_t bytes) {
        if (out_begin_ == (size_t)-1) out_begin_ = top_;
        outs_.push_back({host, top_, bytes});
        return bump(bytes, out_end_);
    }
    size_t scratch(size_t bytes) { size_t dummy; return bump(bytes, dummy); }
    template <class T = void> T *ptr(size_t off) const { return (T *)(a_->dev + off); }

    hipError_t upload() {
        hipError_t e;
        if ((e = a_->reserve(top_ ? top_ : 256)) != hipSuccess || (e = a_->init_pin()) != hipSuccess) return e;
        dirty_ = true;
        if (in_end_ <= Arena::PIN_BYTES) {
            for (const Item &it : ins_) std::memcpy(a_->pin + it.off, it.host, it.bytes);
            return in_end_ ? hipMemcpyAsync(a_->dev, a_->pin, in_end_, hipMemcpyHostToDevice, st_) : hipSuccess;
        }
        size_t k = 0;   // running chunk counter over all items: the halves alternate across item boundaries too
        for (const Item &it : ins_) {
            if (it.bytes >= ((size_t)1 << 20) && host_pinned(it.host)) {   // page-locked caller memory: one direct DMA
                if ((e = hipMemcpyAsync(a_->dev + it.off, it.host, it.bytes, hipMemcpyHostToDevice, st_)) != hipSuccess) return e;
                continue;
            }
            for (size_t done = 0; done < it.bytes; done += Arena::HALF, ++k) {
                const size_t n = it.bytes - done < Arena::HALF ? it.bytes - done : Arena::HALF;
                char *half = a_->pin + (k & 1) * Arena::HALF;
                if (k >= 2 && (e = hipEventSynchronize(a_->ev[k & 1])) != hipSuccess) return e;
                CopyPool::get().copy(half, (const char *)it.host + done, n);
                if ((e = hipMemcpyAsync(a_->dev + it.off + done, half, n, hipMemcpyHostToDevice, st_)) != hipSuccess) return e;
                if ((e = hipEventRecord(a_->ev[k & 1], st_)) != hipSuccess) return e;
            }
        }
        return hipSuccess;
    }

    // More device work follows a download() (second round of a call): the arena is busy again until the next download().
    void touch() { dirty_ = true; }

    // Copies every out() buffer back and synchronises the stream.
    hipError_t download() {
        hipError_t e;
        const size_t span = out_begin_ == (size_t)-1 ? 0 : out_end_ - out_begin_;
        if (span <= Arena::PIN_BYTES) {
            if (span && (e = hipMemcpyAsync(a_->pin, a_->dev + out_begin_, span, hipMemcpyDeviceToHost, st_)) != hipSuccess) return e;
            if ((e = hipStreamSynchronize(st_)) != hipSuccess) return e;
            dirty_ = false;
            for (const Item &it : outs_)
                if (it.host) std::memcpy(it.host, a_->pin + (it.off - out_begin_), it.bytes);
            return hipSuccess;
        }
        // the uploads' last two chunks may still own the halves
        if ((e = hipStreamSynchronize(st_)) != hipSuccess) return e;
        struct Chunk { const Item *it; size_t done, n; };
        std::vector<Chunk> ch;
        bool direct = false;
        for (const Item &it : outs_) {
            if (!it.host) continue;
            if (it.bytes >= ((size_t)1 << 20) && host_pinned(it.host)) {   // page-locked caller memory: one direct DMA
                if ((e = hipMemcpyAsync(it.host, a_->dev + it.off, it.bytes, hipMemcpyDeviceToHost, st_)) != hipSuccess) return e;
                direct = true;
                continue;
            }
            for (size_t done = 0; done < it.bytes; done += Arena::HALF)
                ch.push_back({&it, done, it.bytes - done < Arena::HALF ? it.bytes - done : Arena::HALF});
        }
        auto issue = [&](size_t k) -> hipError_t {
            hipError_t e2 = hipMemcpyAsync(a_->pin + (k & 1) * Arena::HALF, a_->dev + ch[k].it->off + ch[k].done, ch[k].n,
                                           hipMemcpyDeviceToHost, st_);
            return e2 != hipSuccess ? e2 : hipEventRecord(a_->ev[k & 1], st_);
        };
        if (!ch.empty() && (e = issue(0)) != hipSuccess) return e;
        for (size_t k = 0; k < ch.size(); ++k) {
            if (k + 1 < ch.size() && (e = issue(k + 1)) != hipSuccess) return e;   // DMA of the next half overlaps this copy
            if ((e = hipEventSynchronize(a_->ev[k & 1])) != hipSuccess) return e;
            CopyPool::get().copy((char *)ch[k].it->host + ch[k].done, a_->pin + (k & 1) * Arena::HALF, ch[k].n);
        }
        if (direct && (e = hipStreamSynchronize(st_)) != hipSuccess) return e;
        dirty_ = false;
        return hipSuccess;
    }

private:
    struct Item { void *host; size_t off, bytes; };
    size_t bump(size_t bytes, size_t &end) {
        const size_t off = top_;
        top_ = (top_ + bytes + 255) & ~(size_t)255;
        end = off + bytes;
        return off;
    }
    Arena *a_;
    hipStream_t st_;
    std::vector<Item> ins_, outs_;
    size_t top_ = 0, in_end_ = 0, out_begin_ = (size_t)-1, out_end_ = 0;
    bool dirty_ = false;
};

}  // namespace csp
