// minsnap_twist.hip -- dispatcher of the lane-pair sweep's four variants (minsnap_twist_impl.h)
#include "minsnap_twist_launch.h"

namespace csp {
namespace twist {

hipError_t launch_twist_f32(const GenericArgs &, const int32_t *, const int64_t *, MixedTable *, double *, size_t, int, hipStream_t);
hipError_t launch_twist_f32s(const GenericArgs &, const int32_t *, const int64_t *, MixedTable *, double *, size_t, int, hipStream_t);
hipError_t launch_twist_f64(const GenericArgs &, const int32_t *, const int64_t *, MixedTable *, double *, size_t, int, hipStream_t);
hipError_t launch_twist_f64s(const GenericArgs &, const int32_t *, const int64_t *, MixedTable *, double *, size_t, int, hipStream_t);

hipError_t launch_twist(const GenericArgs &a, bool f32, const int32_t *perm, const int64_t *coef_off, MixedTable *tab, double *ckws,
                        size_t ck_role_doubles, int workgroups, hipStream_t st) {
    if (a.status) return f32 ? launch_twist_f32s(a, perm, coef_off, tab, ckws, ck_role_doubles, workgroups, st)
                             : launch_twist_f64s(a, perm, coef_off, tab, ckws, ck_role_doubles, workgroups, st);
    return f32 ? launch_twist_f32(a, perm, coef_off, tab, ckws, ck_role_doubles, workgroups, st)
               : launch_twist_f64(a, perm, coef_off, tab, ckws, ck_role_doubles, workgroups, st);
}

}  // namespace twist
}  // namespace csp
