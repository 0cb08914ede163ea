// minsnap_fixedpath_o3.hip -- instantiates the register-resident path-penalty kernels
// (minsnap_fixed_path_impl.h) for derivative order 3, S = 2..16 segments.
#include "minsnap_fixed_path_impl.h"

namespace csp {

hipError_t launch_fixedpath_o3(const GenericArgs &a, hipStream_t st) {
    switch (a.S) {
        case 2: return fixedk::launch_path_s<3, 2>(a, st);
        case 3: return fixedk::launch_path_s<3, 3>(a, st);
        case 4: return fixedk::launch_path_s<3, 4>(a, st);
        case 5: return fixedk::launch_path_s<3, 5>(a, st);
        case 6: return fixedk::launch_path_s<3, 6>(a, st);
        case 7: return fixedk::launch_path_s<3, 7>(a, st);
        case 8: return fixedk::launch_path_s<3, 8>(a, st);
        case 9: return fixedk::launch_path_s<3, 9>(a, st);
        case 10: return fixedk::launch_path_s<3, 10>(a, st);
        case 11: return fixedk::launch_path_s<3, 11>(a, st);
        case 12: return fixedk::launch_path_s<3, 12>(a, st);
        case 13: return fixedk::launch_path_s<3, 13>(a, st);
        case 14: return fixedk::launch_path_s<3, 14>(a, st);
        case 15: return fixedk::launch_path_s<3, 15>(a, st);
        case 16: return fixedk::launch_path_s<3, 16>(a, st);
    }
    return hipErrorInvalidValue;
}

}  // namespace csp
