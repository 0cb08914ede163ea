// minsnap_fixedpath_o4b.hip -- instantiates the register-resident path-penalty kernels
// (minsnap_fixed_path_impl.h) for derivative order 4, S = 10..16 segments.
#include "minsnap_fixed_path_impl.h"

namespace csp {

hipError_t launch_fixedpath_o4b(const GenericArgs &a, hipStream_t st) {
    switch (a.S) {
        case 10: return fixedk::launch_path_s<4, 10>(a, st);
        case 11: return fixedk::launch_path_s<4, 11>(a, st);
        case 12: return fixedk::launch_path_s<4, 12>(a, st);
        case 13: return fixedk::launch_path_s<4, 13>(a, st);
        case 14: return fixedk::launch_path_s<4, 14>(a, st);
        case 15: return fixedk::launch_path_s<4, 15>(a, st);
        case 16: return fixedk::launch_path_s<4, 16>(a, st);
    }
    return hipErrorInvalidValue;
}

}  // namespace csp
