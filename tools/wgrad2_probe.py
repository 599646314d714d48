"""Driver for a rocprofv3 pass over the two-hidden-layer weight-gradient kernel (csrc/mlp_grad.hip: mlp2_wgrad_kernel, the three
H x H products per row on v_mfma_f32_16x16x4_f32), at the row counts of the C* workload.

    python3 tools/wgrad2_probe.py                                   (HIP events: time and TFLOP/s per shape)
    rocprofv3 --kernel-trace --stats --output-format csv -d out -- python3 tools/wgrad2_probe.py
    rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES \
              --output-format csv -d out2 -- python3 tools/wgrad2_probe.py
Algorithmic flop per row: 2 H (Din + Dout) for the outer layers' forward, twice that for their reverse, and 3 x 2 H^2 for the
hidden-to-hidden layer (pre-activations, d h1, dWh)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from psvo_amd import ops  # noqa: E402

# (shape of the row tensor (T, B, Din, N[, M]), Din, H, Dout): the backward simulation's rows and the filter's rows at C*
SHAPES = [((200, 32, 2, 128, 16), 2, 64, 2), ((200, 32, 2, 128, 16), 2, 64, 1), ((200, 32, 2, 128), 2, 64, 2),
          ((200, 32, 2, 128, 16), 2, 32, 2), ((1000, 8, 4, 512, 16), 4, 64, 4)]


def main():
    # PSVO_WGRAD2=2|3 (read by psvo_amd._lib): the bf16-split variants of the H x H products
    print("PSVO_WGRAD2 =", os.environ.get("PSVO_WGRAD2", "0 (f32 matrix instruction)"))
    g = torch.Generator().manual_seed(0)
    for shape, Din, H, Dout in SHAPES:
        dshape = shape[:2] + (Dout,) + shape[3:]
        X = torch.randn(*shape, generator=g).cuda()
        dO = torch.randn(*dshape, generator=g).cuda()
        w = (torch.randn(Din, H, generator=g) / Din ** 0.5, torch.zeros(H), torch.randn(H, Dout, generator=g) / H ** 0.5,
             torch.zeros(Dout), torch.randn(H, H, generator=g) / H ** 0.5, torch.zeros(H))
        w = tuple(t.cuda() for t in w)
        ops.mlp_wgrad(X, dO, w, Din, H, Dout)
        torch.cuda.synchronize()
        n = 10
        e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        e[0].record()
        for _ in range(n):
            ops.mlp_wgrad(X, dO, w, Din, H, Dout)
        e[1].record()
        torch.cuda.synchronize()
        t = e[0].elapsed_time(e[1]) / n * 1e-3
        rows = X.numel() // Din
        flop = rows * (6.0 * H * H + 6.0 * H * (Din + Dout))
        print("rows=%d Din=%d H=%d Dout=%d  %.3f ms = %.1f TFLOP/s (%.0f %% of the 157.3 f32 MFMA peak)"
              % (rows, Din, H, Dout, t * 1e3, flop / t / 1e12, 100.0 * flop / t / 157.3e12))


if __name__ == "__main__":
    main()
