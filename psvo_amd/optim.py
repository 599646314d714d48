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
        # Layout: the tensors of every MLP with one or two hidden layers first, as [W1 | b1 | (Wh | bh |) W2 | b2] (the layout the native
        # weight-gradient kernels write; nn.Module registration order would put mu_kernel / mu_bias first), then
        # everything else in registration order.  Shared parameters (f == q1 under use_bootstrap) appear once.
        from .transformation.MLP import MLP_transformation
        self.params, seen = [], set()

        def add(q):
            if q.requires_grad and id(q) not in seen:
                seen.add(id(q))
                self.params.append(q)
        for mod in module.modules():
            if isinstance(mod, MLP_transformation) and len(mod.Dhs) in (1, 2):
                for q in self._mlp_order(mod):
                    add(q)
        for q in module.parameters():
            add(q)
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
        self._annotate(module)

    @staticmethod
    def _mlp_order(mod):
        """[W1 | b1 | W2 | b2] (psvo_mlp_wgrad) or, two hidden layers, [W1 | b1 | Wh | bh | W2 | b2] (psvo_mlp2_wgrad)"""
        ps = []
        for W, b in zip(mod.kernels, mod.biases):
            ps += [W, b]
        return ps + [mod.mu_kernel, mod.mu_bias]

    def _annotate(self, module):
        """Tell the native backward passes where each module's gradient lives in the flat buffer so they can
        accumulate into it directly (one launch) instead of returning tensors for autograd to add one by one:
          * MLP_transformation with one hidden layer: [W1 | b1 | W2 | b2] -- exactly psvo_mlp_wgrad's layout; with two:
            [W1 | b1 | Wh | bh | W2 | b2] -- psvo_mlp2_wgrad's;
          * LSTMBlockCellLayer: [kernel | bias];
          * all tf_mvn.sigma_con vectors, when contiguous: one fused softplus/clamp for every distribution."""
        from .distribution.mvn import tf_mvn
        from .model import LSTMBlockCellLayer
        from .transformation.MLP import MLP_transformation
        off, o = {}, 0
        for q in self.params:
            off[id(q)] = o
            o += q.numel()

        def block(ps):
            if not all(id(q) in off for q in ps):
                return None
            start = off[id(ps[0])]
            o = start
            for q in ps:
                if off[id(q)] != o:
                    return None
                o += q.numel()
            return start, o

        for mod in module.modules():
            mod.__dict__.pop("_flat_grad", None)
            if isinstance(mod, MLP_transformation) and len(mod.Dhs) in (1, 2):
                b = block(self._mlp_order(mod))
            elif isinstance(mod, LSTMBlockCellLayer):
                b = block([mod.kernel, mod.bias])
            else:
                continue
            if b is not None:
                mod.__dict__["_flat_grad"] = self.grad[b[0]:b[1]]
        dists = []
        for mod in module.modules():
            if isinstance(mod, tf_mvn) and all(mod is not d for d in dists):
                dists.append(mod)
        dists.sort(key=lambda d: off.get(id(d.sigma_con), -1))
        b = block([d.sigma_con for d in dists]) if dists else None
        module.__dict__["_sigma_block"] = None
        if b is not None:
            mins = torch.cat([torch.full((d.sigma_con.numel(),), float(d.sigma_min)) for d in dists]).to(self.flat.device)
            module.__dict__["_sigma_block"] = {"raw": self.flat[b[0]:b[1]], "grad": self.grad[b[0]:b[1]], "mins": mins,
                                               "dists": dists, "sizes": [d.sigma_con.numel() for d in dists]}

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
