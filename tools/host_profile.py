"""Host-side issue time of a training step (no device sync inside the loop): how far ahead of the GPU the launching
thread runs, split into forward issue / backward issue / optimizer issue.

    python tools/host_profile.py [workload]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from psvo_amd.optim import FlatParams, TFAdam  # noqa: E402

wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "C*"]
FLAGS, model, smc = bench.build_objective(wl, "cuda")
smc.generator = torch.Generator(device="cuda").manual_seed(0)
hidden, obs = bench.fhn_batch(wl[1], wl[2], 100, "cuda")
flat = FlatParams(model)
opt = TFAdam(flat)


def step(t):
    t0 = time.perf_counter()
    flat.zero_grad()
    z, _ = smc.get_log_ZSMC(obs, hidden)
    t1 = time.perf_counter()
    z.backward()
    t2 = time.perf_counter()
    opt.step(3e-3)
    t3 = time.perf_counter()
    t[0] += t1 - t0; t[1] += t2 - t1; t[2] += t3 - t2


for _ in range(10):
    step([0, 0, 0])
torch.cuda.synchronize()
n, t = 50, [0.0, 0.0, 0.0]
w0 = time.perf_counter()
for _ in range(n):
    step(t)
w1 = time.perf_counter()
torch.cuda.synchronize()
w2 = time.perf_counter()
print("host issue per step: forward %.3f ms, backward %.3f ms, optimizer %.3f ms; loop %.3f ms/step; "
      "with final sync %.3f ms/step" % (t[0] / n * 1e3, t[1] / n * 1e3, t[2] / n * 1e3, (w1 - w0) / n * 1e3,
                                        (w2 - w0) / n * 1e3))
