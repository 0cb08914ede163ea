/*
 * csp_bezier.h -- C-ABI of the batched Bezier path sampler (second half: "batched Bezier
 * sampler", math_util/bezier.cpp:28-190).  Not a solver: per-segment cubic Bezier evaluation, embarrassingly
 * parallel over paths and segments.
 *
 * Replaces math_util::Bezier::GenerateTrajectoryMatrix (math_util/bezier.hpp:109, math_util/bezier.cpp:127-190; per
 * segment Init + GeneratePath, :18-118) for a batch of paths:
 *   - headings: one-sided differences at the ends, central differences inside (:144-159);
 *   - control points: p1 = p0 + (cos, sin)(heading0) * chord * k, p2 = p3 - (cos, sin)(heading3) * chord * k, z by thirds;
 *     k starts at 1/3 and grows by 0.02 (at most 10 tries, capped at 0.45) until the curvature |v x a| / |v|^3 at
 *     t in {0, 0.5, 1} respects 1/min_radius -- skipped for min_radius <= 1 (:40-94);
 *   - samples: t = 0, t += resolution / (|p2 - p1|_xy + 2/3 chord) while t <= 1, the parameter ACCUMULATED like the
 *     reference's loop (:105-116); every segment after a path's first drops its first sample (:169-171);
 *   - a segment whose end points are closer than 0.1 in the plane contributes its end point only (:37, :174-178).
 * Same conventions as csp_minsnap.h: plain pointers, caller-owned buffers, csp_status codes, HIP on gfx950 only,
 * no CPU fallback.
 *
 * Layouts (fp64, row-major): waypoints [total_points][3] -- the paths concatenated; path b owns points
 * offsets[b] .. offsets[b+1]-1 (a path of fewer than 2 points yields no samples, :129-131);
 * samples [batch][capacity][3]; counts [batch] = the TRUE number of samples of each path (rows beyond `capacity`
 * are dropped: counts[b] > capacity tells the caller to retry with a larger capacity).
 */
#ifndef CSP_BEZIER_H_
#define CSP_BEZIER_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* resolution: the reference's `sample_distance_override` when > 0, otherwise 1.0 (:133-136) -- pass the effective value.
 * min_radius: BezierConfig::min_radius (default 1.0 = no curvature constraint; Bezier_3D sets 300, uavPathPlanning.cpp:4492-4494).
 * mem_space: CSP_MEM_HOST (0) stages through the device synchronously, CSP_MEM_DEVICE (1) enqueues on hip_stream. */
int csp_bezier_generate_batch(const double *waypoints, const int64_t *offsets, int64_t batch, double resolution,
                              double min_radius, int64_t capacity, double *samples, int32_t *counts,
                              uint32_t mem_space, int32_t device_id, void *hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* CSP_BEZIER_H_ */
