#!/usr/bin/env python3
"""Timeline of a REPLAYED training step (hipGraph): a one-thread stamp launch (psvo_debug_stamp: device wall clock, 100 MHz)
is captured right before and right after every native launch, on the stream that launch goes to, and at the start and the
end of the step on the capturing stream.  Each stamp costs a launch of its own on its stream, so the step runs a few per cent
longer than the uninstrumented one (printed beside it); the ORDER and the gaps are what this is for.

    python tools/replay_timeline.py [workload] [out.txt]          (on the GPU box)"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(workload="C*", dst=None, detail=True):
    import torch
    import bench
    from psvo_amd import _lib, ops, autograd as _ag
    from psvo_amd.graph import GraphedStep
    from psvo_amd.optim import FlatParams, TFAdam
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    wl = bench.WORKLOADS[workload]
    obj, B, T, N, Dx, Dy, M, H, Dh = wl
    FLAGS, model, smc = bench.build_objective(wl, dev, seed=0)
    smc.generator = torch.Generator(device=dev).manual_seed(1234)
    hidden, obs = bench.fhn_batch(B, T, seed=100, device=dev)
    flat = FlatParams(model)
    opt = TFAdam(flat)
    lib = _lib.load()
    slots = torch.zeros(4096, dtype=torch.int64, device=dev)
    names = []

    def stamp(label, stream):
        i = len(names)
        names.append(label)
        _lib.check(lib.psvo_debug_stamp(ctypes.c_void_p(slots[i:].data_ptr()), ctypes.c_void_p(stream.cuda_stream)),
                   "psvo_debug_stamp")

    def hook(name, phase, stream):
        stamp((name, phase), stream)

    def local_step():
        if hooked[0]:
            stamp(("step", 0), torch.cuda.current_stream())
        flat.zero_grad()
        z, _ = smc.get_log_ZSMC(obs, hidden)
        with _ag.deferred_join():
            z.backward()
        if hooked[0]:
            stamp(("step", 1), torch.cuda.current_stream())
        return z.detach()

    def run(g, n=30):
        for _ in range(5):
            g(); opt.step(3e-3, world_size=1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            g(); opt.step(3e-3, world_size=1)
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / n

    hooked = [False]
    plain = GraphedStep(local_step, generators=[smc.generator])
    ms_plain = run(plain)
    # the warm-up passes of GraphedStep would register their own stamps: count only the captured pass
    hooked[0] = True
    ops.set_timing_hook(hook if detail else None)

    class Counting(GraphedStep):
        pass
    orig_fn = local_step

    def fn():
        del names[:]
        return orig_fn()
    inst = GraphedStep(fn, generators=[smc.generator])
    ops.set_timing_hook(None)
    ms_inst = run(inst)
    inst()
    torch.cuda.synchronize()
    v = slots[:len(names)].cpu().tolist()
    t0 = v[0]
    us = lambda x: (x - t0) / 100.0
    lines = ["# replayed %s training step: %.3f ms uninstrumented, %.3f ms with %d stamps" % (workload, ms_plain, ms_inst, len(names)),
             "# start_us  end_us   dur_us  launch"]
    open_ = {}
    for (name, phase), x in zip(names, v):
        if name == "step":
            lines.append("%9.1f %8s %8s  step %s" % (us(x), "", "", "begins" if phase == 0 else "ends (capturing stream)"))
        elif phase == 0:
            open_[name] = x
        else:
            a = open_.pop(name)
            lines.append("%9.1f %8.1f %8.1f  %s" % (us(a), us(x), (x - a) / 100.0, name))
    text = "\n".join(lines)
    print(text)
    if dst:
        open(dst, "w").write(text + "\n")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "C*", sys.argv[2] if len(sys.argv) > 2 else None)
