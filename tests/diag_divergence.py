"""Diagnostic (test infrastructure, imports the oracle): train the notebook configuration on the GPU step by step, watch the
per-step ELBO / gradient norm, and when a step's ELBO or gradient leaves the plausible range, replay THAT parameter state
through the HIP path and the fp64 oracle with identical injected noise and compare ELBO and every gradient.

    python tests/diag_divergence.py --seed 3 --epochs 10 --out gpurun_out/curve/diag_seed3.json
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import psvo_oracle as O          # noqa: E402
from tests import helpers as Hh              # noqa: E402
from tests import notebook_curve as NC       # noqa: E402
from tests import test_gpu_parity as TP      # noqa: E402


def compare_at_state(model, FLAGS, obs_seq, seed):
    """HIP vs fp64 oracle at the model's current parameters on one sequence with injected noise (free-running draws in the
    oracle, teacher-forced into the HIP run)"""
    from psvo_amd.SMC.PSVO import PSVO
    obs = torch.tensor(obs_seq[None]).double()
    fl = Hh.oracle_flags(FLAGS, "PSVO")
    noise = O.make_noise(fl, 1, obs.shape[1], seed=seed)
    _, ref0 = Hh.run_oracle(model, FLAGS, "PSVO", obs, noise)
    teacher = {"idx_f": ref0["idx_f"], "idx_b": ref0["idx_b"]}
    z_ref, P = TP._oracle_grads(model, FLAGS, "PSVO", obs, noise, teacher)
    nz = Hh.noise_to_hip({**noise, **teacher}, "cuda")
    for key in ("u_f", "u_b"):
        nz.pop(key, None)
    smc = PSVO(model, FLAGS)
    grads_before = [None if p.grad is None else p.grad.clone() for p in model.parameters()]
    for p in model.parameters():
        if p.grad is not None:
            p.grad.zero_()
    z, _ = smc.get_log_ZSMC(obs.float().cuda(), None, noise=nz)
    z.backward()
    torch.cuda.synchronize()
    rows = []
    for name, p, ref in TP._pairs(model, P):
        g = torch.zeros_like(ref) if p.grad is None else p.grad.detach().double().cpu()
        r = torch.zeros_like(ref) if ref.grad is None else ref.grad
        rows.append((name, float((g - r).abs().max()), float(r.abs().max()), float(p.detach().abs().max())))
    for p, g in zip(model.parameters(), grads_before):
        if g is not None:
            p.grad.copy_(g)
    return float(z.detach()), float(z_ref), rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=3)
    ap.add_argument("--epochs", type=int, default=10)
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    from psvo_amd import dp
    from psvo_amd.model import SSM
    from psvo_amd.optim import FlatParams, TFAdam
    from psvo_amd.SMC.PSVO import PSVO
    from psvo_amd.trainer import trainer
    d = NC.fixture()
    FLAGS = NC.notebook_flags(a.seed, a.epochs)
    torch.cuda.set_device(0)
    device = torch.device("cuda", 0)
    dp.init(device=device)
    torch.manual_seed(FLAGS.seed)
    np.random.seed(FLAGS.seed)
    model = SSM(FLAGS).to(device)
    smc = PSVO(model, FLAGS)
    smc.generator = torch.Generator(device=device).manual_seed(FLAGS.seed)
    tr = trainer(model, smc, FLAGS)
    tr.flat = FlatParams(model)
    tr.optimizer = TFAdam(tr.flat)
    obs_train = d["Ytrain"].astype(np.float64)
    hid = np.zeros((len(obs_train), 200, 2))
    order = np.arange(len(obs_train))
    log, report = [], {"seed": a.seed, "events": []}
    step = 0
    prev = tr.flat.flat.clone()
    done = False
    for ep in range(a.epochs):
        order = order[np.random.permutation(len(order))]
        for j in order:
            prev.copy_(tr.flat.flat)
            z = float(tr.train_step(obs_train[j:j + 1], hid[j:j + 1], FLAGS.lr))
            gn = float(tr.flat.grad.norm())
            pm = float(tr.flat.flat.abs().max())
            log.append((step, ep, int(j), z, gn, pm))
            bad = (not np.isfinite(z)) or z < -3000 or gn > 1e5
            if bad and len(report["events"]) < 3:
                # replay the state BEFORE this step
                cur = tr.flat.flat.clone()
                tr.flat.flat.copy_(prev)
                zh, zo, rows = compare_at_state(model, FLAGS, obs_train[j].astype(np.float32), seed=900 + step)
                tr.flat.flat.copy_(cur)
                os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
                torch.save({"flat": prev.cpu(), "obs": obs_train[j].astype(np.float32), "noise_seed": 900 + step,
                            "names": [n for n, _ in model.named_parameters()], "rows": rows},
                           a.out.replace(".json", "_state%d.pt" % len(report["events"])))
                worst = sorted(rows, key=lambda r: -(r[1] / max(r[2], 1e-9)))[:6]
                report["events"].append({"step": step, "epoch": ep, "seq": int(j), "elbo_step": z, "grad_norm": gn,
                                         "replay_elbo_hip": zh, "replay_elbo_oracle": zo, "worst_rel_grad_err": worst,
                                         "param_absmax": sorted(rows, key=lambda r: -r[3])[:5]})
                print(json.dumps(report["events"][-1]), flush=True)
            if not np.isfinite(z) or z < -1e8:
                done = True
                break
            step += 1
        print("epoch", ep + 1, "last elbo", log[-1][3], "max grad norm this epoch",
              max(r[4] for r in log if r[1] == ep), "param absmax", log[-1][5], flush=True)
        if done:
            break
    report["log_tail"] = log[-400:]
    report["grad_norm_quantiles"] = np.quantile([r[4] for r in log], [0.5, 0.9, 0.99, 1.0]).tolist()
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(report, f)


if __name__ == "__main__":
    main()
