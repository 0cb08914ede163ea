// minsnap_fixed_o2.hip -- instantiates the register-resident fixed-size kernels
// (minsnap_fixed_impl.h) for derivative order 2 (polynomial degree 3), S = 2..16 segments.
#include "minsnap_fixed_impl.h"

namespace csp {

hipError_t launch_fixed_o2(const GenericArgs &a, int cus, hipStream_t st) {
    switch (a.S) {
        case 2: return fixedk::launch_s<2, 2, false>(a, cus, st);
        case 3: return fixedk::launch_s<2, 3, false>(a, cus, st);
        case 4: return fixedk::launch_s<2, 4, false>(a, cus, st);
        case 5: return fixedk::launch_s<2, 5, false>(a, cus, st);
        case 6: return fixedk::launch_s<2, 6, false>(a, cus, st);
        case 7: return fixedk::launch_s<2, 7, false>(a, cus, st);
        case 8: return fixedk::launch_s<2, 8, false>(a, cus, st);
        case 9: return fixedk::launch_s<2, 9, false>(a, cus, st);
        case 10: return fixedk::launch_s<2, 10, false>(a, cus, st);
        case 11: return fixedk::launch_s<2, 11, false>(a, cus, st);
        case 12: return fixedk::launch_s<2, 12, false>(a, cus, st);
        case 13: return fixedk::launch_s<2, 13, false>(a, cus, st);
        case 14: return fixedk::launch_s<2, 14, false>(a, cus, st);
        case 15: return fixedk::launch_s<2, 15, false>(a, cus, st);
        case 16: return fixedk::launch_s<2, 16, false>(a, cus, st);
    }
    return hipErrorInvalidValue;
}

}  // namespace csp
