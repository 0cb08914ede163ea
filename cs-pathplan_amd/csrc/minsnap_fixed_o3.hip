// minsnap_fixed_o3.hip -- instantiates the register-resident fixed-size kernels
// (minsnap_fixed_impl.h) for derivative order 3 (polynomial degree 5), S = 2..16.
#include "minsnap_fixed_impl.h"

namespace csp {

hipError_t launch_fixed_o3(const GenericArgs &a, int cus, hipStream_t st) {
    switch (a.S) {
        case 2: return fixedk::launch_hs<3, 1, false>(a, cus, st);
        case 4: return fixedk::launch_hs<3, 2, false>(a, cus, st);
        case 6: return fixedk::launch_hs<3, 3, false>(a, cus, st);
        case 8: return fixedk::launch_hs<3, 4, false>(a, cus, st);
        case 10: return fixedk::launch_hs<3, 5, false>(a, cus, st);
        case 12: return fixedk::launch_hs<3, 6, false>(a, cus, st);
        case 14: return fixedk::launch_hs<3, 7, false>(a, cus, st);
        case 16: return fixedk::launch_hs<3, 8, false>(a, cus, st);
    }
    return hipErrorInvalidValue;
}

}  // namespace csp
