#!/usr/bin/env python3
"""hipGraph capture of one local training step (objective + reverse pass) under different wirings, one subprocess per arm:
a capture that leaves a stream unjoined ends in hipErrorStreamCaptureUnjoined or, when that stream is the null stream,
takes the process down inside hipStreamEndCapture (ROCm 7.2).

    python tools/capture_probe.py            all arms, one line each
    python tools/capture_probe.py <arm>      one arm in this process

flat = parameters and gradients in one buffer (optim.FlatParams: the kernels accumulate weight gradients in place);
noflat = gradients return through autograd; padded = per-particle MLPs narrower than the kernel width (SVO._kernel_width).
Round 2 finding: with output tensors stored on ctx (reference cycle) the previous evaluation's graph survived until the
cycle collector ran, its gradient accumulators were reused together with the stream they were created on, and every
noflat arm failed; see autograd._aliases."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARMS = {
    "aesmc+noflat": dict(obj="AESMC", extra={}, overlap=0, flat=0),
    "aesmc+flat": dict(obj="AESMC", extra={}, overlap=0, flat=1),
    "svo+noflat": dict(obj="SVO", extra={}, overlap=0, flat=0),
    "psvo+nooverlap+noflat": dict(extra={}, overlap=0, flat=0),
    "psvo+overlap+noflat": dict(extra={}, overlap=1, flat=0),
    "psvo+overlap+flat": dict(extra={}, overlap=1, flat=1),
    "psvo+padded+overlap+flat": dict(extra=dict(q1_layers="24", g_layers="16"), overlap=1, flat=1),
    "psvo+padded+overlap+noflat": dict(extra=dict(q1_layers="24", g_layers="16"), overlap=1, flat=0),
    "psvowr+overlap+noflat": dict(obj="PSVOwR", extra={}, overlap=1, flat=0),
    # the trainer's wiring: streams joined by the caller after backward() (autograd.deferred_join)
    "psvo+deferred": dict(extra={}, overlap=1, flat=1, deferred=1),
    # ... with the filter's MLP_g weight gradient on the second side stream (PSVO_FILTER_WGRAD_SPLIT)
    "psvo+deferred+gsplit": dict(extra={}, overlap=1, flat=1, deferred=1, gsplit=1),
}
# NOT part of run_all(): the round-2 abort reproduced on purpose.  Two streams forked into one capture that wait on EACH
# OTHER's events (side waits for the backward simulation's weight gradients on side2; side2 waits for the filter's reverse
# kernel on side) are a DAG of work but a cycle in the runtime's fork relation, and hip::Stream::EndCapture (ROCm 7.2) walks
# that relation recursively without a visited set: stack overflow, SIGSEGV.  `python tools/capture_probe.py psvo+cross-wait`
NEGATIVE = {"psvo+cross-wait": dict(extra={}, overlap=1, flat=1, deferred=1, gsplit=1, cross=1)}


def arm(name):
    sys.path.insert(0, ROOT)
    import torch
    from tests import test_gpu_parity as TP
    from tests import helpers as Hh
    from psvo_amd import autograd
    from psvo_amd.graph import GraphedStep
    from psvo_amd.optim import FlatParams
    a = ARMS.get(name) or NEGATIVE[name]
    autograd.OVERLAP = bool(a["overlap"])
    autograd.SPLIT_FILTER_WGRAD = bool(a.get("gsplit"))
    autograd.SPLIT_CROSS_WAIT = bool(a.get("cross"))
    obj = a.get("obj", "PSVO")
    FLAGS, model, smc, obs, noise = TP._setup(obj, 2, 8, 16, 8 if obj.startswith("PSVO") else 1, 2, 1, 32, True, True,
                                              seed=4, **a["extra"])
    nz = Hh.noise_to_hip(noise, "cuda")
    flat = FlatParams(model) if a["flat"] else None
    obs_c = obs.float().cuda()

    def local():
        if flat is not None:
            flat.zero_grad()
        else:
            model.zero_grad(set_to_none=False)
        z, _ = smc.get_log_ZSMC(obs_c, None, noise=nz)
        if a.get("deferred"):
            with autograd.deferred_join():
                z.backward()
        else:
            z.backward()
        return z.detach()
    z_e = local().clone()            # eager, on the null stream
    torch.cuda.synchronize()
    step = GraphedStep(local)
    z_g = step()
    torch.cuda.synchronize()
    assert torch.equal(z_e, z_g), (float(z_e), float(z_g))
    print(name, "ok", float(z_e), float(z_g), flush=True)


def run_all(names=None):
    bad = []
    for n in (names or ARMS):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), n], stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, text=True, timeout=300)
        tail = (r.stdout.strip().splitlines() or [""])[-1]
        print("%-28s rc=%d  %s" % (n, r.returncode, tail), flush=True)
        if r.returncode:
            bad.append((n, r.returncode, r.stderr[-400:]))
    return bad


if __name__ == "__main__":
    if len(sys.argv) > 1:
        arm(sys.argv[1])
    else:
        sys.exit(1 if run_all() else 0)
