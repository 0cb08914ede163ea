// minsnap_twist_f64.hip -- the lane-pair sweep of the mixed-order entry (minsnap_twist_impl.h), double storage,
// per-trajectory status off; one translation unit per variant so that they compile in parallel.
#include "minsnap_twist_impl.h"

CSP_TWIST_INSTANTIATE(double, false, launch_twist_f64)
