// Explicit instantiation of the backward-simulation reverse kernels for Dx = 4 (see bsim_bwd_impl.h).
#include "bsim_bwd_impl.h"

namespace psvo {
template int bb_dispatch_dy<4>(const BsimBwdArgs&, const BsimBwdOut&, int, int, int, hipStream_t);
}  // namespace psvo
