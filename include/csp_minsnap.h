/*
 * csp_minsnap.h -- C-ABI of the MI355X-native batched minimum-snap solver.
 *
 * This is the drop-in boundary for ONE hot path of MEZHANGYUE/CS-PathPlan: the closed-form
 * minimum-snap QP behind `TrajectoryGeneratorTool` (reference interface:
 * math_util/minimum_snap.hpp:36-63; implementation math_util/minimum_snap.cpp:22-649).
 * The reference is a C++ class with Eigen types in its signatures; a C++ shim with the same
 * class surface (cs-pathplan_amd/host/math_util/minimum_snap.hpp) forwards to the entry points below, so
 * `UavPathPlanner::Minisnap_3D/Minisnap_EN` (uavPathPlanning.cpp:4401-4474) compile unchanged.
 * See INTEGRATION.md for the binding a reference maintainer would add.
 *
 * Conventions shared by every entry point
 *   - plain pointers and sizes only; caller owns every buffer; nothing throws across the ABI;
 *   - return value: CSP_OK (0) or a negative csp_status; per-trajectory problems are reported
 *     through the optional `status` array, never by aborting the batch;
 *   - silent: nothing is printed (the reference prints inside the solver);
 *   - thread-safe for distinct streams; no global mutable state besides the lazily built
 *     constant tables and the pool of staging arenas host-memory calls borrow from (mutex-protected);
 *   - the compute path is HIP on gfx950 only.  There is NO CPU fallback: without a usable
 *     device the calls return CSP_ERR_NO_DEVICE.
 *
 * Data layout (all row-major, contiguous; device pointers 16-byte aligned -- hipMalloc and torch
 * allocations are -- or pass CSP_FLAG_FORCE_GENERIC)
 *   waypoints : [B][S+1][3]   positions (reference `Path`, W x 3, minimum_snap.hpp:47)
 *   times     : [B][S]        segment durations (reference `Time`, minimum_snap.hpp:50)
 *   bc        : [B or 1][4][3] rows = start vel, end vel, start acc, end acc
 *                             (reference `Vel` 2x3 and `Acc` 2x3, minimum_snap.hpp:48-49)
 *   coeffs    : [B][S][3][2*order]  polynomial coefficients per segment and axis, HIGHEST power
 *               first, local time t in [0, T_seg] -- the row-major image of the reference's
 *               PolyCoeff (S x 3*p_num1d, minimum_snap.cpp:220-223, :626-648)
 *   ragged batches (num_segments == 0): trajectories are concatenated; trajectory b owns
 *               segments seg_offsets[b] .. seg_offsets[b+1]-1 of `times`/`coeffs` and waypoints
 *               seg_offsets[b]+b .. seg_offsets[b+1]+b of `waypoints`.
 */
#ifndef CSP_MINSNAP_H_
#define CSP_MINSNAP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CSP_MINSNAP_ABI_VERSION 1u

typedef enum csp_status {
    CSP_OK = 0,
    CSP_ERR_INVALID_ARG = -1, /* null pointer, bad order/segment count, bad abi_version      */
    CSP_ERR_UNSUPPORTED = -2, /* order outside 1..5 (the reference's int arithmetic overflows
                                 from order 6, minimum_snap.cpp:321-323)                      */
    CSP_ERR_WORKSPACE = -3,   /* workspace missing or smaller than csp_minsnap_workspace_bytes */
    CSP_ERR_HIP = -4,         /* a HIP runtime call failed; see csp_minsnap_last_hip_error    */
    CSP_ERR_NO_DEVICE = -5    /* no gfx950 device visible -- there is no CPU fallback         */
} csp_status;

enum { CSP_DTYPE_F64 = 0, CSP_DTYPE_F32 = 1 };
enum { CSP_MEM_HOST = 0, CSP_MEM_DEVICE = 1 };

/* flags */
#define CSP_FLAG_FORCE_GENERIC 0x1u /* never dispatch the register-resident fixed-size kernel */
#define CSP_FLAG_SEGMENT_MAJOR 0x2u /* uniform batches only: coeffs laid out [S][B][3][2*order]
                                       (segment-major) instead of [B][S][3][2*order]; each
                                       (segment, trajectory) record keeps the reference's row
                                       content (3*p_num1d values, highest power first) */

#define CSP_FLAG_F32_ARITH 0x8u      /* CSP_DTYPE_F32 only: compute in fp32 too.  By default fp32 is
                                       the STORAGE type and the arithmetic is fp64 (pure fp32 loses
                                       3..5 digits at order 4..5) */
#define CSP_FLAG_LONG_SEGMENTS 0x10u /* csp_minsnap_sample_batch with device memory: segments hold hundreds of
                                       0.1-s candidates each (long legs): sample with one wave per trajectory.
                                       Host-memory calls decide this themselves from the times. */
#define CSP_FLAG_SPAN 0x20u          /* 16 < S <= 256: use the 16-segments-per-lane kernel (the default beyond 256
                                       segments) instead of the 4-segments-per-lane one (A/B testing) */
#define CSP_FLAG_NO_PERSISTENT 0x4u  /* fixed kernel: one workgroup per 64 trajectories instead of
                                       persistent workgroups with LDS-DMA prefetch (A/B testing) */

/* per-trajectory status bits written to `status` */
#define CSP_TRAJ_OK 0
#define CSP_TRAJ_NONFINITE 1   /* a coefficient is inf/NaN (the reference would return it silently).  The register-resident
                                  kernels (uniform S <= 16, with or without the path penalty) test the highest-power and
                                  the constant coefficient of every record -- every input and unknown of the segment
                                  enters the former, the latter is the start waypoint --, so an inf/NaN that ARISES in a
                                  middle coefficient alone (finite inputs of magnitude >~ 1e300) is not flagged there; the
                                  chunked, span and generic kernels test every stored coefficient                       */
#define CSP_TRAJ_NOT_SPD 2     /* a pivot of the free-derivative Hessian R_PP was <= 0              */
#define CSP_TRAJ_SKIPPED 4     /* csp_minsnap_solve_mixed only: the trajectory was NOT solved (its order is outside 2..5
                                  or its segment count outside 1..min(256, max_segments)); its coefficient block is left untouched (zero-filled with CSP_MEM_HOST) */

typedef struct csp_minsnap_desc {
    uint32_t abi_version;       /* CSP_MINSNAP_ABI_VERSION                                       */
    uint32_t dtype;             /* CSP_DTYPE_F64 | CSP_DTYPE_F32: STORAGE type of waypoints/times/bc/coeffs */
    int32_t order;              /* derivative order d_order (reference `order`): 4 = min-snap,
                                   polynomial degree 2*order-1 (minimum_snap.cpp:237-238)         */
    int32_t num_segments;       /* uniform S >= 1, or 0 for a ragged batch                         */
    int64_t batch;              /* B >= 0                                                          */
    const int64_t *seg_offsets; /* ragged only: [B+1] prefix sums, same memory space as the data   */
    int32_t max_segments;       /* ragged only: max_b S_b (sizes the workspace)                    */
    uint32_t bc_per_trajectory; /* 0: bc is [1][4][3] shared by the batch; 1: [B][4][3]            */
    double path_weight;         /* reference `path_weight` (minimum_snap.cpp:233, :347-469)        */
    double vel_zero_weight;     /* reference `vel_zero_weight` (:234, :473-509)                    */
    const double *vel_zero_weight_per_traj; /* optional [B] (always f64) overriding the scalar;
                                   used by the batched re-solve loop (minimum_snap.cpp:80-90)      */
    uint32_t mem_space;         /* CSP_MEM_HOST: pointers are host memory, the call stages through
                                   the device and is synchronous; CSP_MEM_DEVICE: device pointers,
                                   the call only enqueues work on `hip_stream`                     */
    int32_t device_id;          /* HIP device ordinal; -1 = current device                         */
    uint32_t flags;
    uint32_t reserved;
} csp_minsnap_desc;

/* Replaces TrajectoryGeneratorTool::SolveQPClosedForm (math_util/minimum_snap.hpp:45-53,
 * minimum_snap.cpp:227-649) for a batch of independent trajectories.
 *   max_dev : optional [B] f64, the reference's *max_deviation out-parameter per trajectory
 *   status  : optional [B] i32, CSP_TRAJ_* bits
 *   workspace / workspace_bytes : device scratch of at least csp_minsnap_workspace_bytes(desc)
 *             bytes (CSP_MEM_DEVICE); may be NULL/0 with CSP_MEM_HOST (carved from the cached arena)
 *   hip_stream : hipStream_t (NULL = default stream) */
int csp_minsnap_solve_batch(const csp_minsnap_desc *desc, const void *waypoints, const void *times,
                            const void *bc, void *coeffs, double *max_dev, int32_t *status,
                            void *workspace, size_t workspace_bytes, void *hip_stream);

/* Device scratch bytes the call above needs for `desc` (0 when the fixed-size kernel serves it). */
size_t csp_minsnap_workspace_bytes(const csp_minsnap_desc *desc);

/* The same solve spread over `ngpu` devices of this node from ONE process (the reference planner
 * is a single C++ process).  Trajectories are independent
 * (minimum_snap.cpp has no cross-trajectory term), so the batch is cut into `ngpu` contiguous shards, shard g on the
 * g-th gfx950 device; no collective other than the scatter of inputs and the gather of results.  Synchronous.
 * ngpu <= 0 uses every gfx950 device; ngpu greater than the device count is CSP_ERR_INVALID_ARG.
 *   CSP_MEM_HOST   : every shard is staged from / to the caller's host memory by a host thread of its own through that
 *                    device's cached arena (page-locked halves, DMA overlapped with the host copy); desc->device_id is
 *                    ignored.  PCIe-bound (1.3-1.55e7 solves/s measured on one device).
 *   CSP_MEM_DEVICE : the batch is RESIDENT on a root device (desc->device_id, or the current one), uniform batches only
 *                    (ragged: CSP_ERR_UNSUPPORTED).  Inputs are scattered and coefficients gathered over RCCL / xGMI from
 *                    that root -- single-process communicators (ncclCommInitAll, cached), grouped ncclSend / ncclRecv,
 *                    one compute and one communication stream per device; every shard is cut into chunks (4; at least
 *                    4096 trajectories each; CSP_SHARD_CHUNKS overrides) and the gather of chunk i overlaps the solve of
 *                    chunk i+1.  The root's own shard is solved in place.  The call first synchronises the root device
 *                    (it takes no stream).
 * The span-versus-chunked kernel choice is made ONCE from the whole batch and pinned for the shards / chunks; within the
 * fixed-size kernels the narrow small-batch variants are bit-equal with the wide ones (tests), so results equal
 * csp_minsnap_solve_batch on one device bit for bit.  UNVERIFIED ON MORE THAN ONE GPU: the development boxes have one
 * device -- what runs there is ngpu = 1 (both forms), the chunk / peer schedule over a recording transport on the CPU
 * (tests/test_shard_schedule.py), and tests that run only where two devices exist. */
int csp_minsnap_solve_batch_sharded(const csp_minsnap_desc *desc, const void *waypoints, const void *times,
                                    const void *bc, void *coeffs, double *max_dev, int32_t *status, int ngpu);

/* n INDEPENDENT uniform batches of one shape (same order, segment count, dtype, weights, flags: `desc`; desc->batch is
 * ignored) with their own buffers, in ONE call -- and, for the shapes of the register-resident fixed-size kernels without
 * the path penalty, in ONE kernel launch: a launch costs ~3 us of dispatch / first-load / end-of-kernel latency around the
 * 3 us of work of a 4096-trajectory batch (BASELINE config 2), so a planner that solves many small batches per tick pays
 * mostly launches.  The workgroups find their (batch, slice) in a table passed as a kernel argument (no upload).
 * The reference solves one flight per call (TrajectoryGeneratorTool::SolveQPClosedForm, math_util/minimum_snap.hpp:45-53).
 *   batches[k]            : trajectories of batch k
 *   waypoints/times/bc/coeffs[k] : batch k's buffers, layouts as csp_minsnap_solve_batch (bc [1][4][3] or [B_k][4][3])
 *   status                : NULL, or n pointers to [B_k] int32 arrays (all or none)
 * CSP_MEM_DEVICE only (CSP_ERR_UNSUPPORTED otherwise, as for ragged descriptors and per-trajectory weights); other shapes
 * are served by one ordinary launch per batch when they need no workspace.  Results equal csp_minsnap_solve_batch's per
 * batch bit for bit.  Asynchronous on `hip_stream`. */
int csp_minsnap_solve_multi(const csp_minsnap_desc *desc, int n, const int64_t *batches, const void *const *waypoints,
                            const void *const *times, const void *const *bc, void *const *coeffs, int32_t *const *status,
                            void *hip_stream);

/* Mixed-ORDER ragged batches in ONE call (BASELINE config 5: per-trajectory segment count AND derivative order).  The
 * reference solves one flight per call with one `order` (TrajectoryGeneratorTool::SolveQPClosedForm,
 * math_util/minimum_snap.hpp:45-53); a planner that batches flights of different smoothness classes would otherwise have to
 * sort them by order itself.  Here the bucketing by (order, length class) runs on the device and the solve reads the inputs
 * and writes the coefficients IN THE CALLER'S ORDER -- nothing is gathered or un-permuted:
 *   desc        : dtype, batch, seg_offsets ([B+1], ragged layout as above), max_segments (<= 256), bc_per_trajectory,
 *                 vel_zero_weight(_per_traj), mem_space, device_id as for csp_minsnap_solve_batch; desc->order and
 *                 num_segments are ignored; path_weight must be 0 and CSP_FLAG_F32_ARITH is not offered (CSP_ERR_UNSUPPORTED)
 *   orders      : [B] int32, derivative order of every trajectory, 2..5 (same memory space as the data)
 *   coeffs      : trajectory b's block [S_b][3][2*order_b] starts at element coeff_offsets[b] = sum_{k<b} E_k, concatenated in
 *                 caller order, where E_k = 6*order_k*S_k rounded up to a whole number of 16-byte pieces (fp32 storage: a
 *                 multiple of 4 elements, i.e. 2 floats of padding after a block of odd order AND odd segment count; fp64:
 *                 no padding ever) -- every block starts 16-byte aligned.  The caller sizes the array (sum_b E_b <= the
 *                 tight sum + 2*B elements), e.g. from its own host copy of the shapes
 *   coeff_offsets_out : optional [B+1] int64, the offsets above (computed on the device); [B] = the total
 *   status      : optional [B] int32 CSP_TRAJ_* bits; trajectories outside the served range get CSP_TRAJ_SKIPPED
 *   workspace   : >= csp_minsnap_mixed_workspace_bytes(desc) bytes of device memory (CSP_MEM_DEVICE); NULL/0 with CSP_MEM_HOST
 *                 (it holds the bucketing tables and the checkpoint slots of the solve below: up to 0.2 MB per persistent
 *                 workgroup at max_segments = 64, 103 MB from B = 16384 on; less for shorter trajectories)
 * Trajectories of up to 64 segments are solved by a sequential twisted block-LDL^T sweep, two lanes per trajectory, in
 * blocks of segments whose factors are recomputed from checkpoints (cs-pathplan_amd/csrc/minsnap_twist_impl.h: the arithmetic
 * of the register-resident fixed-size kernels), every order in ONE persistent launch; longer ones (65..256 segments) by
 * csp_minsnap_solve_batch's workspace-free chunked kernel, one launch per order.  A trajectory longer than
 * desc->max_segments is reported CSP_TRAJ_SKIPPED.
 * CSP_MEM_DEVICE: asynchronous on `hip_stream`.  CSP_MEM_HOST: staged through the cached arena, synchronous. */
int csp_minsnap_solve_mixed(const csp_minsnap_desc *desc, const int32_t *orders, const void *waypoints, const void *times,
                            const void *bc, void *coeffs, int64_t *coeff_offsets_out, int32_t *status,
                            void *workspace, size_t workspace_bytes, void *hip_stream);
size_t csp_minsnap_mixed_workspace_bytes(const csp_minsnap_desc *desc);

/* Replaces the time-allocation step of TrajectoryGeneratorTool::GenerateTrajectoryMatrix
 * (minimum_snap.cpp:59-72): T_i = max(|p_{i+1}-p_i| / V_avg, min_time_s), or min_time_s when
 * V_avg <= 1e-6.  Same layouts / descriptor as the solve (path/vel weights ignored). */
int csp_minsnap_time_alloc_batch(const csp_minsnap_desc *desc, const void *waypoints, double v_avg,
                                 double min_time_s, void *times, void *hip_stream);

/* Replaces the solver half of TrajectoryGeneratorTool::GenerateTrajectoryMatrix
 * (minimum_snap.cpp:59-90) for a batch: time allocation (as csp_minsnap_time_alloc_batch) followed
 * by the re-solve loop -- solve; while max_deviation > 0.2 and fewer than 10 increases:
 * vel_zero_weight <- (w < 1e-6 ? 0.01 : 2w), solve again -- tracked per trajectory on the device
 * (no host round trip: converged trajectories are skipped by the later passes).
 *   times      : out, [B][S] (same layout rules as the solve)
 *   coeffs     : out
 *   max_dev    : out, optional [B] f64 (final deviation metric)
 *   vel_zero_weight_out : optional [B] f64, the weight the final solve used.  max_dev, vel_zero_weight_out and
 *                iterations are LIVE LOOP STATE during the call (the passes read and update them in place); with
 *                CSP_MEM_DEVICE vel_zero_weight_out may alias desc->vel_zero_weight_per_traj (weights updated in place)
 *   iterations : optional [B] i32, number of weight increases (reference `iter`)
 *   workspace  : >= csp_minsnap_plan_workspace_bytes(desc) bytes (device); NULL/0 with CSP_MEM_HOST */
int csp_minsnap_plan_batch(const csp_minsnap_desc *desc, const void *waypoints, double v_avg, double min_time_s,
                           const void *bc, void *times, void *coeffs, double *max_dev,
                           double *vel_zero_weight_out, int32_t *iterations, int32_t *status,
                           void *workspace, size_t workspace_bytes, void *hip_stream);
size_t csp_minsnap_plan_workspace_bytes(const csp_minsnap_desc *desc);

/* Replaces the sampling half of GenerateTrajectoryMatrix (minimum_snap.cpp:97-205): evaluates each
 * trajectory at dt = min(0.1, T_seg/10), keeps a point whenever it is >= sample_distance away from
 * the previously kept one, appends the end point, and computes the two statistics the reference
 * prints (max climb/descent rate, min turn radius).
 *   samples : out, [B][capacity][3] (storage dtype); trajectory b uses the first counts[b] rows
 *   counts  : out, [B] i32 -- the TRUE number of samples; rows beyond `capacity` are dropped, so
 *             counts[b] > capacity tells the caller to retry with a larger capacity
 *   stats   : optional out, [B][2] f64 = {max climb rate, min turn radius} */
int csp_minsnap_sample_batch(const csp_minsnap_desc *desc, const void *times, const void *coeffs,
                             double sample_distance, int64_t capacity, void *samples, int32_t *counts,
                             double *stats, void *hip_stream);

/* Replaces the WHOLE of TrajectoryGeneratorTool::GenerateTrajectoryMatrix (minimum_snap.cpp:22-206) for a batch:
 * csp_minsnap_plan_batch followed by csp_minsnap_sample_batch with the times and coefficients left on the device.
 * Results are bit for bit those of the two calls.  A CSP_MEM_HOST caller -- the reference's own call pattern, one
 * flight per call (uavPathPlanning.cpp:4423, :4461) -- pays ONE upload, ONE download and ONE synchronisation (a
 * second round only when the re-solve loop had to raise somebody's weight).
 *   capacity : rows of `samples` per trajectory (csp_minsnap_sample_capacity gives a sufficient value)
 *   times, coeffs : optional outputs with CSP_MEM_HOST; REQUIRED (device buffers, they are the intermediates) with
 *                   CSP_MEM_DEVICE
 *   workspace : >= csp_minsnap_plan_workspace_bytes(desc) bytes (device); NULL/0 with CSP_MEM_HOST
 * Everything else as in the two calls it stands for. */
int csp_minsnap_generate_batch(const csp_minsnap_desc *desc, const void *waypoints, double v_avg, double min_time_s,
                               const void *bc, double sample_distance, int64_t capacity, void *samples,
                               int32_t *counts, double *stats, void *times, void *coeffs, double *max_dev,
                               double *vel_zero_weight_out, int32_t *iterations, int32_t *status,
                               void *workspace, size_t workspace_bytes, void *hip_stream);

/* Upper bound of the samples any trajectory of the batch can produce (every candidate of the sampling loop,
 * minimum_snap.cpp:126-161, plus the first and the last point), from HOST-resident waypoints: a `capacity` that
 * csp_minsnap_generate_batch / csp_minsnap_sample_batch cannot overflow.  -1 for an invalid descriptor. */
int64_t csp_minsnap_sample_capacity(const csp_minsnap_desc *desc, const void *waypoints_host, double v_avg,
                                    double min_time_s);

/* Name of the kernel csp_minsnap_solve_batch would dispatch for `desc` ("fixed_o4_s16_f64",
 * "generic_o4_f64", ...); NULL for an invalid descriptor.  For tests and profiles. */
const char *csp_minsnap_kernel_name(const csp_minsnap_desc *desc);

/* Number of visible HIP devices whose architecture is gfx950 (0 => every solve call fails). */
int csp_minsnap_device_count(void);

/* Process-lifetime notes: the library keeps helper threads (staging copies), cached arenas, streams and RCCL communicators
 * alive once used -- do not dlclose() it or fork() after the first call; HIP device ordinals up to 63 are supported.
 * An idle arena keeps at most 512 MB of device memory (CSP_ARENA_KEEP_MB), larger ones are freed when the call returns. */
/* CSP_MEM_HOST calls stage through per-device arenas (one device allocation + 16 MB of page-locked memory each) that are
 * cached between calls, so that the reference's call pattern -- one flight per call, uavPathPlanning.cpp:4423/:4461 -- does
 * not pay hipMalloc/hipFree every time.  This frees the arenas no call is using (optional; e.g. before a long idle phase). */
void csp_minsnap_release_cached_memory(void);

const char *csp_minsnap_version(void);
const char *csp_minsnap_strerror(int status);
const char *csp_minsnap_last_hip_error(void);

#ifdef __cplusplus
}
#endif
#endif /* CSP_MINSNAP_H_ */
