// Two-hidden-layer build of filter_fwd.hip: per-particle MLPs relu(relu(x W1 + b1) Wh + bh) W2 + b2 (psvo_desc.layers == 2;
// reference src/transformation/MLP.py:24-38 with *_layers = "H,H").  See PSVO_L in common.h.
#define PSVO_L 2
#include "filter_fwd.hip"
