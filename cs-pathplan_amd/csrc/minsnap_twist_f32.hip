// minsnap_twist_f32.hip -- the lane-pair sweep of the mixed-order entry (minsnap_twist_impl.h), float storage,
// per-trajectory status off; one translation unit per variant so that they compile in parallel.
#include "minsnap_twist_impl.h"

CSP_TWIST_INSTANTIATE(float, false, launch_twist_f32)
