// minsnap_fixed_o5.hip -- instantiates the register-resident fixed-size kernels
// (minsnap_fixed_impl.h) for derivative order 5 (polynomial degree 9), S = 2..8 segments.
#include "minsnap_fixed_impl.h"

namespace csp {

hipError_t launch_fixed_o5(const GenericArgs &a, int cus, hipStream_t st) {
    switch (a.S) {
        case 2: return fixedk::launch_s<5, 2, false>(a, cus, st);
        case 3: return fixedk::launch_s<5, 3, false>(a, cus, st);
        case 4: return fixedk::launch_s<5, 4, false>(a, cus, st);
        case 5: return fixedk::launch_s<5, 5, false>(a, cus, st);
        case 6: return fixedk::launch_s<5, 6, false>(a, cus, st);
        case 7: return fixedk::launch_s<5, 7, false>(a, cus, st);
        case 8: return fixedk::launch_s<5, 8, false>(a, cus, st);
    }
    return hipErrorInvalidValue;
}

}  // namespace csp
