// minsnap_twist_f64s.hip -- the lane-pair sweep of the mixed-order entry (minsnap_twist_impl.h), double storage,
// per-trajectory status on; one translation unit per variant so that they compile in parallel.
#include "minsnap_twist_impl.h"

CSP_TWIST_INSTANTIATE(double, true, launch_twist_f64s)
