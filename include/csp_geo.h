/*
 * csp_geo.h -- C-ABI of the batched WGS84 <-> ENU transforms: the step either side of the
 * minimum-snap path in the reference planner.
 *
 * Replaces UavPathPlanner::wgs84ToENU_Batch / enuToWGS84_Batch
 * (uavPathPlanning.cpp:1085-1108; per-point bodies :1046-1083, building blocks :893-1043,
 * constants uavPathPlanning.hpp:133-173).  Same conventions as csp_minsnap.h: plain pointers,
 * caller-owned buffers, status codes (csp_status), HIP on gfx950 only, no CPU fallback.
 *
 * Layouts (fp64, row-major): lla [n][3] = {lon_deg, lat_deg, alt_m} (struct WGS84Point field
 * order), enu [n][3] = {east, north, up} (struct ENUPoint), ref [3] = the reference point as lla
 * (read on the HOST: it is one point, passed by value in the reference).
 */
#ifndef CSP_GEO_H_
#define CSP_GEO_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* mem_space: CSP_MEM_HOST (0) stages through the device synchronously, CSP_MEM_DEVICE (1) enqueues
 * on hip_stream.  device_id -1 = current device. */
int csp_geo_wgs84_to_enu_batch(const double *lla, const double *ref_host, double *enu, int64_t n,
                               uint32_t mem_space, int32_t device_id, void *hip_stream);

/* Inverse (ENU -> ECEF -> WGS84 with the reference's <=10-step fixed-point latitude iteration,
 * tolerance 1e-12 rad, uavPathPlanning.cpp:926-968). */
int csp_geo_enu_to_wgs84_batch(const double *enu, const double *ref_host, double *lla, int64_t n,
                               uint32_t mem_space, int32_t device_id, void *hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* CSP_GEO_H_ */
