"""A/B of the one-hidden-layer weight-gradient kernels (csrc/mlp_grad.hip) at the row counts of the bench workloads:
mlp_wgrad_kernel (round 1, PSVO_WGRAD_OLD=1) against mlp_wgrad_cols_kernel (the default: shared rows, incremental index,
rows requested one iteration ahead).

    python3 tools/wgrad_probe.py
Algorithmic flop per row: 2 H (Din + Dout) forward (pre-activations, d h), the same again for dW1 / dW2, + 2 H for the relu
and db1."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from psvo_amd import ops  # noqa: E402

# (shape of the row tensor (T, B, Din, N[, M]), Din, H, Dout)
SHAPES = [((200, 32, 2, 128, 16), 2, 32, 2),      # C*: MLP_f rows of the backward simulation
          ((200, 32, 2, 128, 16), 2, 32, 1),      # C*: MLP_g rows of the backward simulation
          ((200, 32, 2, 128), 2, 32, 2),          # C*: the filter's rows
          ((400, 32, 3, 128, 16), 3, 32, 3),      # C3
          ((500, 8, 4, 512, 16), 4, 32, 4),       # C5
          ((500, 8, 4, 512, 16), 4, 32, 1),
          ((200, 32, 2, 128, 16), 2, 64, 2)]


def main():
    g = torch.Generator().manual_seed(0)
    for shape, Din, H, Dout in SHAPES:
        dshape = shape[:2] + (Dout,) + shape[3:]
        X = torch.randn(*shape, generator=g).cuda()
        dO = torch.randn(*dshape, generator=g).cuda()
        w = (torch.randn(Din, H, generator=g) / Din ** 0.5, 0.3 * torch.randn(H, generator=g),
             torch.randn(H, Dout, generator=g) / H ** 0.5, torch.zeros(Dout))
        w = tuple(t.cuda() for t in w)
        rows = X.numel() // Din
        flop = rows * (4.0 * H * (Din + Dout) + 2.0 * H)
        line, outs = [], []
        for old in ("1", "0"):
            os.environ["PSVO_WGRAD_OLD"] = old
            outs.append(ops.mlp_wgrad(X, dO, w, Din, H, Dout).clone())
            torch.cuda.synchronize()
            n = 10
            e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            e[0].record()
            for _ in range(n):
                ops.mlp_wgrad(X, dO, w, Din, H, Dout)
            e[1].record()
            torch.cuda.synchronize()
            t = e[0].elapsed_time(e[1]) / n * 1e-3
            line.append("%s %.3f ms (%.1f TFLOP/s)" % ("old" if old == "1" else "new", t * 1e3, flop / t / 1e12))
        print("rows=%d Din=%d H=%d Dout=%d  %s  max rel diff %.1e"
              % (rows, Din, H, Dout, "  ".join(line),
                 float((outs[0] - outs[1]).abs().max() / outs[0].abs().max())), flush=True)


if __name__ == "__main__":
    main()
