// Two-hidden-layer build (psvo_desc.layers == 2) of bsim_bwd_dx2.hip; see PSVO_L in common.h.
#define PSVO_L 2
#include "bsim_bwd_dx2.hip"
