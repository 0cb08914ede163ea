// minsnap_fixed_o4a.hip -- instantiates the register-resident fixed-size kernels
// (minsnap_fixed_impl.h) for derivative order 4 (polynomial degree 7), S = 2..9 segments.
#include "minsnap_fixed_impl.h"

namespace csp {

hipError_t launch_fixed_o4a(const GenericArgs &a, int cus, hipStream_t st) {
    switch (a.S) {
        case 2: return fixedk::launch_s<4, 2, true>(a, cus, st);
        case 3: return fixedk::launch_s<4, 3, true>(a, cus, st);
        case 4: return fixedk::launch_s<4, 4, true>(a, cus, st);
        case 5: return fixedk::launch_s<4, 5, true>(a, cus, st);
        case 6: return fixedk::launch_s<4, 6, true>(a, cus, st);
        case 7: return fixedk::launch_s<4, 7, true>(a, cus, st);
        case 8: return fixedk::launch_s<4, 8, true>(a, cus, st);
        case 9: return fixedk::launch_s<4, 9, true>(a, cus, st);
    }
    return hipErrorInvalidValue;
}

}  // namespace csp
