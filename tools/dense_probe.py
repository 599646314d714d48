"""Driver for a rocprofv3 pass over the Dense-layer kernels (csrc/dense.hip: v_mfma_f32_16x16x4_f32): a few forward /
backward launches at the shapes the hoisted networks of the path produce, plus one large shape for the rate.

    rocprofv3 --kernel-trace --stats --output-format csv -d out -- python3 tools/dense_probe.py
    rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES \
              --output-format csv -d out2 -- python3 tools/dense_probe.py
Prints achieved TFLOP/s per shape from HIP events (2 R Din Dout flop forward; 4 R Din Dout + 2 R Dout backward)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from psvo_amd import ops  # noqa: E402

SHAPES = [(6400, 64, 64), (6400, 256, 64), (6400, 64, 2), (262144, 256, 256)]


def main():
    g = torch.Generator().manual_seed(0)
    for R, Din, Dout in SHAPES:
        X = torch.randn(R, Din, generator=g).cuda()
        W = (torch.randn(Din, Dout, generator=g) / Din ** 0.5).cuda()
        b = torch.zeros(Dout).cuda()
        dY = torch.randn(R, Dout, generator=g).cuda()
        Y = ops.dense_forward(X, W, b, True)
        ops.dense_backward(X, Y, dY, W, True)
        torch.cuda.synchronize()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        n = 20
        e[0].record()
        for _ in range(n):
            Y = ops.dense_forward(X, W, b, True)
        e[1].record()
        for _ in range(n):
            ops.dense_backward(X, Y, dY, W, True)
        e[2].record()
        torch.cuda.synchronize()
        tf, tb = e[0].elapsed_time(e[1]) / n * 1e-3, e[1].elapsed_time(e[2]) / n * 1e-3
        print("R=%d Din=%d Dout=%d  forward %.1f us = %.2f TFLOP/s   backward %.1f us = %.2f TFLOP/s"
              % (R, Din, Dout, tf * 1e6, 2.0 * R * Din * Dout / tf / 1e12, tb * 1e6,
                 (4.0 * R * Din * Dout + 2.0 * R * Dout) / tb / 1e12))


if __name__ == "__main__":
    main()
