// Explicit instantiation of the second-layout backward-simulation reverse kernels for Dx = 2 (see bsim_bwd2_impl.h).
#include "bsim_bwd2_impl.h"

namespace psvo {
template int bb2_dispatch_dy<2>(const BsimBwdArgs&, const BsimBwdOut&, int, int, int, int, hipStream_t);
}  // namespace psvo
