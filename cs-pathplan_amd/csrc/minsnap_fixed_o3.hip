// minsnap_fixed_o3.hip -- instantiates the register-resident fixed-size kernels
// (minsnap_fixed_impl.h) for derivative order 3 (polynomial degree 5), S = 2..16 segments.
#include "minsnap_fixed_impl.h"

namespace csp {

hipError_t launch_fixed_o3(const GenericArgs &a, int cus, hipStream_t st) {
    switch (a.S) {
        case 2: return fixedk::launch_s<3, 2, false>(a, cus, st);
        case 3: return fixedk::launch_s<3, 3, false>(a, cus, st);
        case 4: return fixedk::launch_s<3, 4, false>(a, cus, st);
        case 5: return fixedk::launch_s<3, 5, false>(a, cus, st);
        case 6: return fixedk::launch_s<3, 6, false>(a, cus, st);
        case 7: return fixedk::launch_s<3, 7, false>(a, cus, st);
        case 8: return fixedk::launch_s<3, 8, false>(a, cus, st);
        case 9: return fixedk::launch_s<3, 9, false>(a, cus, st);
        case 10: return fixedk::launch_s<3, 10, false>(a, cus, st);
        case 11: return fixedk::launch_s<3, 11, false>(a, cus, st);
        case 12: return fixedk::launch_s<3, 12, false>(a, cus, st);
        case 13: return fixedk::launch_s<3, 13, false>(a, cus, st);
        case 14: return fixedk::launch_s<3, 14, false>(a, cus, st);
        case 15: return fixedk::launch_s<3, 15, false>(a, cus, st);
        case 16: return fixedk::launch_s<3, 16, false>(a, cus, st);
    }
    return hipErrorInvalidValue;
}

}  // namespace csp
