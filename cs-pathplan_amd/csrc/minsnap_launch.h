// minsnap_launch.h -- host-visible launch interface between the C-ABI (minsnap_capi.hip) and
// the kernels (minsnap_generic.hip, minsnap_fixed.hip).  Internal; the public boundary is
// include/csp_minsnap.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <cstdio>

namespace csp {

// csp_minsnap_solve_multi: up to 32 independent uniform batches of one shape in ONE launch of the fixed-size kernels; a
// workgroup finds its (batch, slice) from first_slice[] -- passed BY VALUE as a kernel argument (no table upload)
struct MultiEntry {
    const void *wp, *tm, *bc;
    void *co;
    int32_t *status;
    int64_t B;
};
struct MultiTable {
    int n = 0;                 // 0: an ordinary single-batch launch
    int first_slice[33];       // first workgroup of batch k; [n] = the grid size
    MultiEntry e[32];
};

struct GenericArgs {
    const void *wp;         // [B][S+1][3] (or ragged concatenation)
    const void *times;      // [B][S]
    const void *bc;         // [B or 1][4][3]
    void *coeffs;           // [B][S][3][2o]
    double *max_dev;        // [B] or null
    int32_t *status;        // [B] or null
    const int64_t *seg_off; // ragged prefix sums or null
    void *ws;               // [(Smax-1)][n*n+3n][B] reals
    int *tstar;             // [Smax][B] (path penalty only) or null
    const double *vw_per;   // [B] or null
    double path_weight;
    double vel_zero_weight;
    int64_t B;
    int S;                  // uniform S (ignored when seg_off != null)
    int order;
    int bc_per_traj;
    int seg_major;          // coeffs laid out [S][B][3][2o] instead of [B][S][3][2o] (uniform S only)
    int64_t Btotal;         // seg_major: the batch size the layout is indexed by
    int64_t Boffset;        // seg_major: first trajectory of this launch inside that batch
    int persistent;         // fixed kernel: persistent workgroups with LDS-DMA prefetch (default on)
    const int32_t *skip;    // generic kernel: [B] non-zero = leave this trajectory untouched (or null)
    int tau_mode;           // path kernel: 0 find t* each call; 1 find and store in tstar; 2 reuse tstar (re-solve loop)
    int slice_w = 64;       // fixed kernels, one-workgroup-per-slice variant: trajectories per workgroup (small batches
                            // are cut into narrower slices so that every CU gets one, see fixedk::narrow_slice)
    int nt_stores = 0;      // fixed kernels, whole slices: non-temporal coefficient stores -- set by the launcher for
                            // batches whose coefficients exceed the Infinity Cache (fixedk::store16)
    int stagger = 0;        // path kernels (orders 3-4): start delay of a CU's second workgroup, units of 8128 clocks
    const MultiTable *multi = nullptr;   // HOST pointer, launcher only: csp_minsnap_solve_multi's batches (then wp/times/bc/coeffs/
                                         // status/B above are ignored)
};

hipError_t launch_generic(const GenericArgs &a, bool f32, bool f32_arith, hipStream_t st);
size_t generic_ws_entries(int order);

// Fixed-size register-resident kernels (minsnap_fixed*.hip): f64, uniform 2 <= S <= 16, orders 2..5,
// zero-velocity penalty everywhere, path penalty for orders 2..4 (bucket table in minsnap_fixed.hip).
bool fixed_supported(int order, int S, bool f32, double path_weight, bool ragged, bool seg_major);
hipError_t launch_fixed(const GenericArgs &a, hipStream_t st);
const char *fixed_kernel_name(int order, int S, bool path);

// Workspace-free multi-lane kernel for ragged / long trajectories (minsnap_chunked.hip): orders 2..5,
// S <= 256, f64 or f32 storage (fp64 arithmetic), zero-velocity penalty, no path penalty.
bool chunked_supported(int order, int Smax, bool f32_arith, double path_weight, bool seg_major);
int chunked_lanes_log2(int Smax);
hipError_t launch_chunked(const GenericArgs &a, bool f32, int Smax, hipStream_t st);

// Mixed-ORDER ragged batches (minsnap_mixed.hip): device-side bucketing by (order, length class), one persistent launch per
// order, inputs read and coefficients written in the caller's order.  a.seg_off, a.B, a.S (= the caller's max_segments),
// a.wp/times/bc/coeffs/status, a.vw_per and the weights are used; `orders` is [B] int32 on the device; `workspace` >= mixed_workspace_bytes(B, a.S); coef_off_out:
// optional [B+1] int64 (element offsets of every trajectory's coefficient block).
size_t mixed_workspace_bytes(int64_t B, int max_segments);
hipError_t launch_mixed(const GenericArgs &a, bool f32, const int32_t *orders, void *workspace, int64_t *coef_off_out, hipStream_t st);

// Long trajectories (16 < S <= 1024): spans of 16 segments per lane, recovery by recomputation
// (minsnap_span.hip); same options as the chunked kernel.
bool span_supported(int order, int Smax, bool f32_arith, double path_weight, bool seg_major);
int span_lanes_log2(int Smax);
hipError_t launch_span(const GenericArgs &a, bool f32, int Smax, hipStream_t st);

struct TimeAllocArgs {
    const void *wp;
    void *times;
    const int64_t *seg_off;
    int64_t B;
    int S;
    double v_avg, min_time_s;
};
hipError_t launch_time_alloc(const TimeAllocArgs &a, bool f32, hipStream_t st);

// Re-solve loop bookkeeping (minimum_snap.cpp:80-90) and polynomial sampling (:97-205), minsnap_plan.hip
hipError_t launch_resolve_init(double *vw, int32_t *iters, int32_t *done, int32_t *pending, double vw0, int64_t B, hipStream_t st);
hipError_t launch_fill_f64(double *p, double v, int64_t n, hipStream_t st);
hipError_t launch_resolve_update(const double *max_dev, double *vw, int32_t *iters, int32_t *done, int32_t *pending, int64_t B,
                                 hipStream_t st);
struct SampleArgs {
    const void *times, *coeffs;
    const int64_t *seg_off;
    void *samples;      // [B][capacity][3]
    int32_t *counts;    // [B]
    double *stats;      // [B][2] max climb rate, min turn radius (or null)
    int64_t B, capacity;
    int S, order, seg_major;
    int Smax;           // longest trajectory (uniform: S)
    int one_lane;       // force the one-lane-per-trajectory kernel (CSP_FLAG_FORCE_GENERIC)
    int long_segments;  // hundreds of candidates per segment: one wave per trajectory (CSP_FLAG_LONG_SEGMENTS,
                        // or found from host-resident times)
    double sample_distance;
    // The reference keeps a candidate when sqrt(d2) >= sample_distance (minimum_snap.cpp:142-150).  sqrt is
    // monotone and correctly rounded, so that is the same as d2 >= keep_dist2 with keep_dist2 = the smallest
    // double whose (IEEE) square root reaches sample_distance -- found on the host once per call, which
    // saves ~27 instructions of fp64 sqrt per candidate.
    double keep_dist2;
};
hipError_t launch_sample(const SampleArgs &a, bool f32, hipStream_t st);
// One WAVE PER SEGMENT for a few long flights (fp64 storage, `capacity` >= every candidate): segments thin independently
// (the reference restarts its reference point at every segment, minimum_snap.cpp:128-133) into per-segment runs of `tmp`
// (tmp_off[g] .. , sized by the caller from the host-resident times, last slot of the trajectory's last segment = p(T)),
// a second kernel places the runs, applies the end-point rule and counts, the statistics kernel follows.
hipError_t launch_sample_segment_waves(const SampleArgs &a, double *tmp, const int64_t *tmp_off, int32_t *seg_counts,
                                       int64_t total_segments, hipStream_t st);

}  // namespace csp
