/*
 * csp_alt.h -- C-ABI of the batched altitude-optimiser solves: the two
 * quadratic programmes the reference hands to Eigen::SimplicialLDLT
 * (UavPathPlanner::optimizeHeights, uavPathPlanning.cpp:1575-1713, solve at :1670-1681;
 *  UavPathPlanner::optimizeHeightsGlobalSmooth, :1715-1827, solve at :1796-1799).
 * Both Hessians are symmetric positive definite and PENTADIAGONAL (second-difference smoothing
 * L^T L + climb-rate first differences + diagonal terms), so the device does a banded LDL^T with
 * two sub-diagonals instead of a general sparse factorisation.
 *
 * Several independent problems (route segments, formation members, Monte-Carlo terrain) are
 * solved per call: problem b owns samples offsets[b] .. offsets[b+1]-1.  Same conventions as
 * csp_minsnap.h (plain pointers, caller-owned buffers, csp_status codes, HIP only).
 * The terrain lookups (cost map / GeoTIFF, uavPathPlanning.cpp:1610-1631) are outside the path:
 * `elev` carries the terrain elevation per sample, NaN where the reference finds none.
 */
#ifndef CSP_ALT_H_
#define CSP_ALT_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct csp_alt_params {   /* UavPathPlanner::AltitudeParams, uavPathPlanning.hpp:415-421 */
    double lambda_smooth;         /* default 1.0  */
    double lambda_follow;         /* default 0.0  */
    double safe_distance;         /* default 50.0 */
    double max_climb_rate;        /* default 2.0  */
} csp_alt_params;

/* Device scratch for `total_points` samples (both entry points). */
size_t csp_alt_workspace_bytes(int64_t total_points);

/* optimizeHeights: xyz [total][3] ENU samples, elev [total], out_z [total].  offsets is int64
 * [batch+1] in the same memory space as the data.  workspace may be NULL with CSP_MEM_HOST. */
int csp_alt_optimize_heights_batch(const double *xyz, const double *elev, const int64_t *offsets, int64_t batch,
                                   const csp_alt_params *params, double *out_z, void *workspace,
                                   size_t workspace_bytes, uint32_t mem_space, int32_t device_id, void *hip_stream);

/* optimizeHeightsGlobalSmooth: fixed end points (weight 1e10), active-set penalty (1e8) for
 * z >= input_z, at most 10 solves per problem, all inside one launch.  solves: optional [batch] i32. */
int csp_alt_global_smooth_batch(const double *input_z, const double *xyz, const int64_t *offsets, int64_t batch,
                                const csp_alt_params *params, double *out_z, int32_t *solves, void *workspace,
                                size_t workspace_bytes, uint32_t mem_space, int32_t device_id, void *hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* CSP_ALT_H_ */
