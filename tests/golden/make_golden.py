"""Generate the golden fixtures committed under tests/golden/.

The reference ships no fixtures for this path and cannot run here (TensorFlow 1.12 / TFP 0.5 are
not installable), so these vectors are produced by the CPU oracle (oracle/psvo_oracle.py, fp64)
in this container -- PARITY UNPINNED w.r.t. the reference itself.  They pin the oracle against
accidental change and give the GPU tests a second, file-based target.

    python tests/golden/make_golden.py [case ...]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import psvo_oracle as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

CASES = {
    # name: (objective, B, T, N, M, Dx, Dy, H, Dh, bootstrap, two_q)
    "psvo_small": ("PSVO", 2, 6, 8, 4, 2, 1, 16, 8, True, True),
    "psvo_lorenz_like": ("PSVO", 2, 5, 12, 8, 3, 1, 16, 8, False, True),
    "aesmc_mid": ("AESMC", 3, 10, 64, 4, 2, 1, 32, 8, True, True),
    "iwae_c1": ("IWAE", 1, 50, 4, 4, 2, 1, 32, 8, True, True),
    "svo_small": ("SVO", 2, 7, 16, 4, 2, 1, 16, 8, True, True),
    "psvowr_small": ("PSVOwR", 2, 6, 12, 4, 2, 1, 16, 8, True, True),
    "psvo_poisson": ("PSVO", 2, 6, 12, 4, 2, 2, 16, 8, True, True),
}
# flags beyond the tuple (same names as the reference's FLAGS)
EXTRA = {"psvo_poisson": dict(poisson_emission=True)}


def flatten(prefix, x, out):
    if torch.is_tensor(x):
        out[prefix] = x.detach().numpy()
    elif isinstance(x, dict):
        for k, v in x.items():
            flatten(prefix + "." + str(k), v, out)
    elif isinstance(x, (list, tuple)):
        for i, v in enumerate(x):
            flatten(prefix + "." + str(i), v, out)
    elif x is None:
        pass
    else:
        out[prefix] = np.asarray(x)


def build(name):
    obj, B, T, N, M, Dx, Dy, H, Dh, boot, two_q = CASES[name]
    fl = dict(Dx=Dx, Dy=Dy, n_particles=N, n_particles_for_BSim_proposal=M, use_bootstrap=boot, use_2_q=two_q,
              objective=obj, layers=[H], y_smoother_Dhs=[Dh], X0_smoother_Dhs=[Dh], sigma_init=1.3, sigma_min=0.5)
    fl.update(EXTRA.get(name, {}))
    P = O.make_params(fl, seed=11, dtype=torch.float64, bias_scale=0.2)
    _, obs = O.fhn_synthetic(B, T, seed=3)
    if Dy != 1 or Dx != 2:
        g = torch.Generator().manual_seed(5)
        obs = torch.randn(B, T, Dy, generator=g, dtype=torch.float64)
    noise = O.make_noise(fl, B, T, seed=77)
    return fl, P, obs, noise


def run(fl, P, obs, noise):
    leaves = []

    def req(x):
        if torch.is_tensor(x):
            x.requires_grad_(True)
            leaves.append(x)
        elif isinstance(x, dict):
            [req(v) for v in x.values()]
        elif isinstance(x, (list, tuple)):
            [req(v) for v in x]
    req(P)
    o = O.OBJECTIVES[fl["objective"]](P, fl)
    z, log = o.get_log_ZSMC(obs, noise)
    z.backward()
    return z, log


def main(names=None):
    for name in (names or CASES):
        fl, P, obs, noise = build(name)
        z, log = run(fl, P, obs, noise)
        out = {"log_ZSMC": z.detach().numpy(), "obs": obs.numpy()}
        flatten("noise", noise, out)
        flatten("params", P, out)
        grads = {}

        def g(prefix, x):
            if torch.is_tensor(x):
                grads[prefix] = (x.grad if x.grad is not None else torch.zeros_like(x)).numpy()
            elif isinstance(x, dict):
                [g(prefix + "." + str(k), v) for k, v in x.items()]
            elif isinstance(x, (list, tuple)):
                [g(prefix + "." + str(i), v) for i, v in enumerate(x)]
        g("grad", P)
        out.update(grads)
        for k in ("Xs", "X_prevs", "X_ancestors", "log_Ws", "idx_f", "bw_Xs", "f_log_probs", "g_log_probs",
                  "bw_log_Omegas", "idx_b", "bw_X_ancestors", "bw_log_W", "idx_r"):
            if log.get(k) is not None:
                out["out." + k] = log[k].detach().numpy()
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, float(z), sum(v.nbytes for v in out.values()) // 1024, "KiB")


if __name__ == "__main__":
    main(sys.argv[1:])       # no arguments: regenerate every fixture
