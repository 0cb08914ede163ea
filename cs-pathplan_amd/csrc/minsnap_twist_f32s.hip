// minsnap_twist_f32s.hip -- the lane-pair sweep of the mixed-order entry (minsnap_twist_impl.h), float storage,
// per-trajectory status on; one translation unit per variant so that they compile in parallel.
#include "minsnap_twist_impl.h"

CSP_TWIST_INSTANTIATE(float, true, launch_twist_f32s)
