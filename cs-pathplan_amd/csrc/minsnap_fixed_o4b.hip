// minsnap_fixed_o4b.hip -- instantiates the register-resident fixed-size kernels
// (minsnap_fixed_impl.h) for derivative order 4 (polynomial degree 7), S = 10..16 segments.
#include "minsnap_fixed_impl.h"

#ifdef CSP_STAMPS
extern "C" int csp_debug_read_stamps(unsigned long long *host, size_t n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(csp_g_stamps), n * sizeof(unsigned long long));
}
#endif

namespace csp {

hipError_t launch_fixed_o4b(const GenericArgs &a, int cus, hipStream_t st) {
    switch (a.S) {
        case 10: return fixedk::launch_s<4, 10, true>(a, cus, st);
        case 11: return fixedk::launch_s<4, 11, true>(a, cus, st);
        case 12: return fixedk::launch_s<4, 12, true>(a, cus, st);
        case 13: return fixedk::launch_s<4, 13, true>(a, cus, st);
        case 14: return fixedk::launch_s<4, 14, true>(a, cus, st);
        case 15: return fixedk::launch_s<4, 15, true>(a, cus, st);
        case 16: return fixedk::launch_s<4, 16, true>(a, cus, st);
    }
    return hipErrorInvalidValue;
}

}  // namespace csp
