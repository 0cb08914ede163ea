// minsnap_fixedpath_o4a.hip -- instantiates the register-resident path-penalty kernels
// (minsnap_fixed_path_impl.h) for derivative order 4, S = 2..9 segments.
#include "minsnap_fixed_path_impl.h"

namespace csp {

hipError_t launch_fixedpath_o4a(const GenericArgs &a, hipStream_t st) {
    switch (a.S) {
        case 2: return fixedk::launch_path_s<4, 2>(a, st);
        case 3: return fixedk::launch_path_s<4, 3>(a, st);
        case 4: return fixedk::launch_path_s<4, 4>(a, st);
        case 5: return fixedk::launch_path_s<4, 5>(a, st);
        case 6: return fixedk::launch_path_s<4, 6>(a, st);
        case 7: return fixedk::launch_path_s<4, 7>(a, st);
        case 8: return fixedk::launch_path_s<4, 8>(a, st);
        case 9: return fixedk::launch_path_s<4, 9>(a, st);
    }
    return hipErrorInvalidValue;
}

}  // namespace csp
