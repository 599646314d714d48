"""Diagnostic (test infrastructure): load a parameter state saved by tests/diag_divergence.py and compare the HIP forward
pass with the fp64 oracle quantity by quantity (teacher-forced indices, identical noise).
    python tests/diag_state.py gpurun_out/curve/diag20_seed3_state0.pt 3"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import psvo_oracle as O          # noqa: E402
from tests import helpers as Hh              # noqa: E402
from tests import notebook_curve as NC       # noqa: E402


def main(path, seed):
    from psvo_amd.model import SSM
    from psvo_amd.optim import FlatParams
    from psvo_amd.SMC.PSVO import PSVO
    st = torch.load(path, weights_only=False)
    FLAGS = NC.notebook_flags(seed, 1)
    torch.manual_seed(seed)
    model = SSM(FLAGS).cuda()
    fp = FlatParams(model)
    fp.flat.copy_(st["flat"].cuda())
    obs = torch.tensor(st["obs"][None]).double()
    fl = Hh.oracle_flags(FLAGS, "PSVO")
    noise = O.make_noise(fl, 1, obs.shape[1], seed=int(st["noise_seed"]))
    z_ref, ref = Hh.run_oracle(model, FLAGS, "PSVO", obs, noise)
    teacher = {"idx_f": ref["idx_f"], "idx_b": ref["idx_b"]}
    nz = Hh.noise_to_hip({**noise, **teacher}, "cuda")
    for key in ("u_f", "u_b"):
        nz.pop(key, None)
    smc = PSVO(model, FLAGS)
    with torch.no_grad():
        z, log = smc.get_log_ZSMC(obs.float().cuda(), None, noise=nz)
    print("ELBO hip %.6g oracle %.6g" % (float(z), float(z_ref)))
    f, b = log["filter"], log["bsim"]
    pairs = [("X", Hh.part_to_ref(f["X"]), ref["X_prevs"]), ("logW", Hh.w_to_ref(f["logW"]), ref["log_Ws"]),
             ("bwX", Hh.part_to_ref(b["bwX"]), ref["bw_Xs"]), ("flp", Hh.w_to_ref(b["flp"]), ref["f_log_probs"]),
             ("glp", Hh.w_to_ref(b["glp"]), ref["g_log_probs"]), ("Omega", Hh.w_to_ref(b["Omega"]), ref["bw_log_Omegas"])]
    for name, h, r in pairs:
        d = (h - r).abs()
        d = d.reshape(d.shape[0], -1).max(1).values
        bad = (d > 1e-2 * (1 + r.abs().reshape(r.shape[0], -1).max(1).values)).nonzero().flatten().tolist()
        print("%-6s max|diff| %.4g  |ref|max %.4g  finite(hip) %s  first bad t: %s" % (
            name, float(d.max()), float(r.abs().max()), bool(torch.isfinite(h).all()), bad[:8]))
        if bad:
            t = bad[-1] if name in ("bwX", "flp", "glp", "Omega") else bad[0]
            print("   t = %d  hip %s\n          ref %s" % (t, h[t].flatten()[:8].tolist(), r[t].flatten()[:8].tolist()))
    print("score hip", log["bsim"]["score"].flatten()[:16].tolist() if "score" in log["bsim"] else None)
    print("sigmas", {k: [round(float(x), 4) for x in v] for k, v in model.sigmas().items()} if hasattr(model, "sigmas") else "")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]))
