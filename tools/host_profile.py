"""cProfile of the host side of C* training steps (which Python frames eat the time between launches)."""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from psvo_amd.optim import FlatParams, TFAdam  # noqa: E402

dev = torch.device("cuda", 0)
wl = bench.WORKLOADS["C*"]
FLAGS, model, smc = bench.build_objective(wl, dev)
smc.generator = torch.Generator(device=dev).manual_seed(0)
hidden, obs = bench.fhn_batch(wl[1], wl[2], 100, dev)
flat = FlatParams(model)
opt = TFAdam(flat)


def step():
    flat.zero_grad()
    z, _ = smc.get_log_ZSMC(obs, hidden)
    z.backward()
    opt.step(3e-3)


for _ in range(5):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(35)

# ---- the autograd engine runs backward nodes on its own thread: time the Python backward bodies directly
import time
from psvo_amd import autograd as A, ops
acc = {}


def wrap(cls, name):
    orig = cls.backward

    def timed(ctx, *g):
        t0 = time.perf_counter()
        out = orig(ctx, *g)
        acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
        return out
    cls.backward = staticmethod(timed)


for cls, name in ((A.FilterFunction, "FilterFunction.backward"), (A.BsimFunction, "BsimFunction.backward"),
                  (A.BiLSTMFunction, "BiLSTMFunction.backward"), (A.SigmaFunction, "SigmaFunction.backward")):
    wrap(cls, name)
for fn in ("bsim_backward", "filter_backward", "mlp_wgrad", "bilstm_backward"):
    o = getattr(ops, fn)

    def mk(o, fn):
        def t(*a, **k):
            t0 = time.perf_counter()
            r = o(*a, **k)
            acc["ops." + fn] = acc.get("ops." + fn, 0.0) + time.perf_counter() - t0
            return r
        return t
    setattr(ops, fn, mk(o, fn))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    step()
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print("per step: issue %.3f ms, to completion %.3f ms" % (t_issue / 20 * 1e3, t_all / 20 * 1e3))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print("  %-28s %.3f ms/step" % (k, v / 20 * 1e3))
