#!/usr/bin/env python3
"""psvo_rows_mlp_backward alone on an idle card: 16 against 64 rows per workgroup (psvo_set_tuning, PSVO_TUNE_ROWS_BWD) over the
row counts and input widths of the hoisted MLPs (q0 / BSim_q_init: B rows; q2 / BSim_q2: B T rows; Din = Dy or 2 Dh).
    python tools/rows_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from psvo_amd import _lib, ops

lib = _lib.load()
H, Dout = 32, 2
print("| rows | Din | 16 rows / workgroup, us | 64 rows / workgroup, us |\n|---|---|---|---|")
for R in (32, 256, 1024, 2048, 6400, 12800, 64000):
    for Din in (1, 64):
        X = torch.randn(R, Din, device="cuda")
        dO = torch.randn(R, Dout, device="cuda")
        w = (torch.randn(Din, H, device="cuda"), torch.randn(H, device="cuda"), torch.randn(H, Dout, device="cuda"),
             torch.randn(Dout, device="cuda"))
        res = []
        for rb in (16, 64):
            assert lib.psvo_set_tuning(_lib.PSVO_TUNE_ROWS_BWD, rb) == 0
            for _ in range(5):
                ops.rows_mlp_backward(X, dO, w)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                ops.rows_mlp_backward(X, dO, w)
            e1.record()
            torch.cuda.synchronize()
            res.append(1e3 * e0.elapsed_time(e1) / 50)
        print("| %d | %d | %.1f | %.1f |" % (R, Din, res[0], res[1]), flush=True)
lib.psvo_set_tuning(_lib.PSVO_TUNE_ROWS_BWD, 0)
