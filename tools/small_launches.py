"""Diagnostic: which PyTorch ops (each one a small kernel on the step's critical streams) a C* training step still issues
around the native launches, by call site.  This is how the unused scale computations of the hoisted means (20 launches
per step) and the zero-filled gradients of the bsim nodes' constant outputs were found.

    python tools/small_launches.py          (GPU box)
"""
import os, sys, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from torch.utils._python_dispatch import TorchDispatchMode
from psvo_amd.optim import FlatParams
wl = bench.WORKLOADS["C*"]
FLAGS, model, smc = bench.build_objective(wl, "cuda")
flat = FlatParams(model)
smc.generator = torch.Generator(device="cuda").manual_seed(0)
hid, obs = bench.fhn_batch(wl[1], wl[2], seed=1, device="cuda")
def step():
    flat.zero_grad(); z, _ = smc.get_log_ZSMC(obs, hid); z.backward(); torch.cuda.synchronize()
for _ in range(3): step()
cnt = collections.Counter()
class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        st = [f for f in traceback.extract_stack() if "psvo_amd" in f.filename]
        where = "%s:%d" % (st[-1].filename.split("psvo_amd/")[-1], st[-1].lineno) if st else "engine"
        cnt[(name, where)] += 1
        return func(*args, **(kwargs or {}))
with Log():
    step()
skip = ("aten.view", "aten.detach", "aten.t.", "aten.transpose", "aten.slice", "aten.select", "aten.permute", "aten.unsqueeze", "aten.expand", "aten.as_strided", "aten._unsafe_view", "aten.reshape", "aten.alias", "aten.split", "aten.squeeze", "aten.empty", "aten.record_stream", "aten.unbind", "aten.lift_fresh", "aten._local_scalar_dense", "aten.is_")
for k, v in sorted(cnt.items(), key=lambda kv: (kv[0][1], kv[0][0])):
    if not any(k[0].startswith(s) for s in skip):
        print(v, k)
