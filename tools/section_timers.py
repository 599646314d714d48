"""Diagnostic: per-section shader-clock cycles of a persistent kernel at a bench workload.

Builds ONE translation unit with -DPSVO_SECTION_TIMERS (plus the product objects of the others) into
psvo_amd/csrc/ab/libpsvo_hip_timers.so, runs a few training steps with that library and prints the
cycles lane 0 of workgroup (0,0) spent between the SEC(i) marks, per kernel launch and per time step.

    python tools/section_timers.py build bsim_fwd bsim_bwd_dx2 filter_fwd filter_bwd     (translation units; build host)
    python tools/section_timers.py run C* bsim_fwd bsim_bwd filter_fwd filter_bwd          (timer names; GPU box)
"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
AB = os.path.join(ROOT, "psvo_amd", "csrc", "ab")
LIBT = os.environ.get("PSVO_TIMERS_LIB") or os.path.join(AB, "libpsvo_hip_timers.so")     # (one unit of a family per library)


def build(units):
    from concurrent.futures import ThreadPoolExecutor
    from psvo_amd import build as B
    B.build_lib(verbose=False)
    os.makedirs(AB, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

    def one(unit):
        obj = os.path.join(AB, unit + "_timers.o")
        subprocess.run([hipcc] + B.FLAGS + ["-DPSVO_SECTION_TIMERS", "-c", os.path.join(B.CSRC, unit + ".hip"),
                        "-o", obj], check=True)
        return obj
    with ThreadPoolExecutor(max_workers=4) as ex:
        tobjs = dict(zip(units, ex.map(one, units)))
    objs = [tobjs.get(s[:-4], os.path.join(B.CSRC, s.replace(".hip", ".o"))) for s in B.SOURCES]
    subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIBT] + objs, check=True)
    print(LIBT)


def legend(unit, lpp):
    """labels of the SEC(i) marks of a timer family, read from the source: `SEC(i);   // text`; the four-lanes-per-particle
    filter kernels mark theirs "(lpp)" (the same timer array serves both kernel shapes, one runs per workload)"""
    import re
    src = {"bsim_bwd2": "bsim_bwd2_impl.h", "bsim_bwd": "bsim_bwd_impl.h"}.get(unit, unit + ".hip")
    out = {}
    for m in re.finditer(r"SEC\((\d+)\);\s*//\s*(.*)", open(os.path.join(ROOT, "psvo_amd", "csrc", src)).read()):
        text = m.group(2).strip()
        is_lpp = text.startswith("(lpp)")
        if unit.startswith("filter") and is_lpp != lpp:
            continue
        out.setdefault(int(m.group(1)), text.replace("(lpp) ", ""))
    return out


def run(workload, units, steps=5):
    os.environ["PSVO_HIP_LIB"] = LIBT
    import torch
    import bench
    from psvo_amd import _lib
    from psvo_amd.optim import FlatParams
    wl = bench.WORKLOADS[workload]
    obj, B, T, N, Dx, Dy, M, H, Dh = wl[:9]
    FLAGS, model, smc = bench.build_objective(wl, "cuda")
    flat = FlatParams(model)
    smc.generator = torch.Generator(device="cuda").manual_seed(0)
    _, obs = bench.fhn_batch(B, T, seed=1, device="cuda")
    lib = ctypes.CDLL(LIBT) if _lib._lib is None else _lib.load()
    fns = {}
    for unit in units:
        fns[unit] = getattr(lib, "psvo_debug_timers_" + unit)
        fns[unit].argtypes = [ctypes.c_void_p, ctypes.c_int]
    buf = (ctypes.c_ulonglong * 32)()

    def step():
        flat.zero_grad() if hasattr(flat, "zero_grad") else None
        z, _ = smc.get_log_ZSMC(obs, None)
        (-z).backward()
        torch.cuda.synchronize()
    for _ in range(3):
        step()
    for unit in units:
        fns[unit](buf, 1)
    for _ in range(steps):
        step()
    for unit in units:
        fns[unit](buf, 0)
        tot = sum(buf)
        print("%s @ %s: cycles of lane 0 / workgroup (0,0), %d launches, T = %d" % (unit, workload, steps, T))
        names = legend(unit, lpp=(N <= 128))
        for i, v in enumerate(buf):
            if v:
                print("  SEC(%2d) %12d  %8.0f cycles/time-step  %5.1f %%   %s"
                      % (i, v, v / steps / T, 100.0 * v / tot, names.get(i, "")))
        print("  total   %12d  %8.0f cycles/time-step" % (tot, tot / steps / T))


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2:])
    else:
        run(sys.argv[2], sys.argv[3:])
