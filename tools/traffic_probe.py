"""Driver for the HBM-traffic PMC passes (run under `rocprofv3 --pmc FETCH_SIZE` and, separately,
`--pmc WRITE_SIZE`): one calibration launch with a known byte count in the kernels' own access
pattern (4-byte-per-lane loads / stores: psvo_adam_step on 32 Mi floats reads 4 arrays and writes 3),
then three C* training steps.  tools/traffic_report.py turns the two CSVs into bytes per launch."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from psvo_amd import _lib  # noqa: E402
from psvo_amd.optim import FlatParams, TFAdam  # noqa: E402

dev = torch.device("cuda", 0)
n = 32 * 1024 * 1024
bufs = [torch.zeros(n, device=dev) for _ in range(4)]
lib = _lib.load()
P = lambda t: ctypes.c_void_p(t.data_ptr())
torch.cuda.synchronize()
lib.psvo_adam_step(P(bufs[0]), P(bufs[1]), P(bufs[2]), P(bufs[3]), n, 1e-3, 0.9, 0.999, 1e-8, 1, 1.0,
                   ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
del bufs

wl = bench.WORKLOADS[os.environ.get("PSVO_WORKLOAD", "C*")]
FLAGS, model, smc = bench.build_objective(wl, dev)
smc.generator = torch.Generator(device=dev).manual_seed(0)
hidden, obs = bench.fhn_batch(wl[1], wl[2], 100, dev)
flat = FlatParams(model)
opt = TFAdam(flat)
for _ in range(3):
    flat.zero_grad()
    z, _ = smc.get_log_ZSMC(obs, hidden)
    z.backward()
    opt.step(3e-3)
torch.cuda.synchronize()
print("probe done", float(z.detach()))
