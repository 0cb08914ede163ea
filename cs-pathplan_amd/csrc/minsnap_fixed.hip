// minsnap_fixed.hip -- register-resident fixed-size kernel (placeholder until the kernel lands).
#include "minsnap_launch.h"
namespace csp {
bool fixed_supported(int, int, bool, double, bool) { return false; }
hipError_t launch_fixed(const GenericArgs &, hipStream_t) { return hipErrorNotSupported; }
const char *fixed_kernel_name(int) { return "fixed_unavailable"; }
}
