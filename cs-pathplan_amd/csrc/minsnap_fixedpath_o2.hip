// minsnap_fixedpath_o2.hip -- instantiates the register-resident path-penalty kernels
// (minsnap_fixed_path_impl.h) for derivative order 2, S = 2..16 segments.
#include "minsnap_fixed_path_impl.h"

#ifdef CSP_STAMPS
extern "C" int csp_debug_read_stamps_path_o2(unsigned long long *host, size_t n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(csp_g_stamps), n * sizeof(unsigned long long));
}
#endif

namespace csp {

hipError_t launch_fixedpath_o2(const GenericArgs &a, hipStream_t st) {
    switch (a.S) {
        case 2: return fixedk::launch_path_s<2, 2>(a, st);
        case 3: return fixedk::launch_path_s<2, 3>(a, st);
        case 4: return fixedk::launch_path_s<2, 4>(a, st);
        case 5: return fixedk::launch_path_s<2, 5>(a, st);
        case 6: return fixedk::launch_path_s<2, 6>(a, st);
        case 7: return fixedk::launch_path_s<2, 7>(a, st);
        case 8: return fixedk::launch_path_s<2, 8>(a, st);
        case 9: return fixedk::launch_path_s<2, 9>(a, st);
        case 10: return fixedk::launch_path_s<2, 10>(a, st);
        case 11: return fixedk::launch_path_s<2, 11>(a, st);
        case 12: return fixedk::launch_path_s<2, 12>(a, st);
        case 13: return fixedk::launch_path_s<2, 13>(a, st);
        case 14: return fixedk::launch_path_s<2, 14>(a, st);
        case 15: return fixedk::launch_path_s<2, 15>(a, st);
        case 16: return fixedk::launch_path_s<2, 16>(a, st);
    }
    return hipErrorInvalidValue;
}

}  // namespace csp
