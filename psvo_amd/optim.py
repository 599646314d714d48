"""Flat parameter storage + the reference's optimizer.

Every trainable variable of the SSM is re-pointed into ONE contiguous fp32 buffer (and its
gradient into one contiguous gradient buffer), so that a training step needs a single RCCL
all-reduce and a single fused Adam launch (psvo_adam_step) whatever the number of variables
(~22.5 k floats for PSVO with H = 32, Dh = 32).
"""
import ctypes

import torch

from . import _lib


class FlatParams(object):
    def __init__(self, module):
        self.params = [p for p in module.parameters() if p.requires_grad]
        # shared parameters (f == q1 under use_bootstrap) appear once in module.parameters()
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.empty(n, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(n, device=dev, dtype=torch.float32)
        off = 0
        for p in self.params:
            k = p.numel()
            self.flat[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + k].view_as(p.data)
            p.grad = self.grad[off:off + k].view_as(p.data)
            off += k
        self.numel = n

    def zero_grad(self):
        self.grad.zero_()


class TFAdam(object):
    """tf.train.AdamOptimizer(lr) (beta1=0.9, beta2=0.999, epsilon=1e-8) on a FlatParams buffer;
    maximises the objective whose gradient sits in flat.grad (the reference minimises -log_ZSMC,
    src/trainer.py:118)."""

    def __init__(self, flat, beta1=0.9, beta2=0.999, epsilon=1e-8):
        self.flat = flat
        self.beta1, self.beta2, self.epsilon = beta1, beta2, epsilon
        self.m = torch.zeros_like(flat.flat)
        self.v = torch.zeros_like(flat.flat)
        self.t = 0

    def step(self, lr, world_size=1):
        """theta <- theta + Adam(d log_ZSMC); flat.grad holds the SUM over ranks of d log_ZSMC."""
        self.t += 1
        lib = _lib.load()
        p = lambda t: ctypes.c_void_p(t.data_ptr())
        st = lib.psvo_adam_step(p(self.flat.flat), p(self.flat.grad), p(self.m), p(self.v), self.flat.numel,
                                float(lr), self.beta1, self.beta2, self.epsilon, self.t, -1.0 / world_size,
                                ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        _lib.check(st, "psvo_adam_step")
