// minsnap_fixed_o4.hip -- instantiates the register-resident fixed-size kernels
// (minsnap_fixed_impl.h) for derivative order 4 (polynomial degree 7), S = 2..16.
#include "minsnap_fixed_impl.h"

#ifdef CSP_STAMPS
extern "C" int csp_debug_read_stamps(unsigned long long *host, size_t n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(csp_g_stamps), n * sizeof(unsigned long long));
}
#endif

namespace csp {

hipError_t launch_fixed_o4(const GenericArgs &a, int cus, hipStream_t st) {
    switch (a.S) {
        case 2: return fixedk::launch_hs<4, 1, true>(a, cus, st);
        case 4: return fixedk::launch_hs<4, 2, true>(a, cus, st);
        case 6: return fixedk::launch_hs<4, 3, true>(a, cus, st);
        case 8: return fixedk::launch_hs<4, 4, true>(a, cus, st);
        case 10: return fixedk::launch_hs<4, 5, true>(a, cus, st);
        case 12: return fixedk::launch_hs<4, 6, true>(a, cus, st);
        case 14: return fixedk::launch_hs<4, 7, true>(a, cus, st);
        case 16: return fixedk::launch_hs<4, 8, true>(a, cus, st);
    }
    return hipErrorInvalidValue;
}

}  // namespace csp
