"""Train the configuration of the reference's notebook and write the learning curve next to the notebook's own log.

The reference published ONE outcome for the path: the training log kept in notebooks/PSVO.ipynb (PSVO, n_particles 16,
n_particles_for_BSim_proposal 8, batch_size 1, T 200, H 32, Dh 32, lr 3e-3, print_freq 10, plateau schedule of
src/trainer.py:244-270, data/fhn/[1,0]_obs_cov_0.01/datadict) -- 40 evaluations of the train / valid log_ZSMC and the
k-step R-square; tests/golden/fhn_notebook.npz holds them with the observations (tests/golden/make_fhn_slice.py).
TF's initial weights and draws cannot be replayed, so the comparison is statistical: seeds differ, the curves must lie on
one another.

    python tests/notebook_curve.py --backend hip    --seeds 0 1 2 --epochs 200 --out profiles/r03_notebook_curve_hip.json
    python tests/notebook_curve.py --backend oracle --seeds 0     --epochs 10  --out profiles/r03_notebook_curve_oracle.json

backend hip     the product: psvo_amd.runner.main(FLAGS) end to end -- loader on a datadict pickle written from the
                fixture, SSM, PSVO, trainer (hipGraph-replayed training step, hand-written reverse pass, native Adam),
                evaluation every print_freq epochs, plateau schedule -- on cuda:0.
backend oracle  TEST INFRASTRUCTURE: the same loop (same initial parameters for a given seed: the SSM is initialised on
                the host either way; same numpy shuffle stream; TF-style Adam restated in torch, src/trainer.py:117) with
                oracle/psvo_oracle.py in fp64 as the per-step compute and torch autograd as its reverse pass, on the CPU.
                It evaluates the two data sets in one batch each (the mean over sequences of the per-sequence estimate is
                the same estimator as the reference's mean over batches of one).
This file lives under tests/ because it imports the oracle."""
import argparse
import json
import math
import os
import pickle
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden", "fhn_notebook.npz")

# the notebook's Experiment_params (notebooks/PSVO.ipynb, output of the tf.app.run() cell)
NOTEBOOK_FLAGS = dict(Dx=2, Dy=1, n_particles=16, n_particles_for_BSim_proposal=8, batch_size=1, lr=3e-3, epoch=400,
                      PSVO=True, print_freq=10, MSE_steps=30, early_stop_patience=200, lr_reduce_patience=30,
                      lr_reduce_factor=1 / math.sqrt(2), min_lr=3e-4, time=200, n_train=200, n_test=40,
                      rslt_dir_name="notebook", save_trajectory=False, save_y_hat=False, saving_num=30)


def notebook_flags(seed, epochs, **kw):
    from psvo_amd.flags import Flags
    f = dict(NOTEBOOK_FLAGS)
    f.update(seed=seed, epoch=epochs)
    f.update(kw)
    return Flags(**f)


def fixture():
    d = np.load(GOLD)
    return {k: d[k] for k in d.files}


def notebook_rows(d, upto=None):
    it = d["nb_iter"]
    keep = it <= (upto if upto is not None else it.max())
    return {"iter": it[keep].tolist(), "train_log_ZSMC": d["nb_train_log_ZSMC"][keep].tolist(),
            "valid_log_ZSMC": d["nb_valid_log_ZSMC"][keep].tolist(),
            "valid_Rsq_k0": d["nb_valid_Rsq"][keep, 0].tolist(), "valid_Rsq_k30": d["nb_valid_Rsq"][keep, 30].tolist(),
            "best_valid_iter": int(d["nb_best_valid_iter"])}


def _history_rows(hist, print_freq):
    n = len(hist["log_ZSMC_tests"])
    it = [1] + [print_freq * k for k in range(1, n)]
    return {"iter": it, "train_log_ZSMC": [float(v) for v in hist["log_ZSMC_trains"]],
            "valid_log_ZSMC": [float(v) for v in hist["log_ZSMC_tests"]],
            "valid_Rsq_k0": [float(r[0]) for r in hist["R_square_tests"]],
            "valid_Rsq_k30": [float(r[-1]) for r in hist["R_square_tests"]],
            "train_Rsq_k0": [float(r[0]) for r in hist["R_square_trains"]]}


# ------------------------------------------------------------------------------------------------------------------
def run_hip(seed, epochs, data=None, quiet=True, **flag_overrides):
    """psvo_amd.runner.main on the fixture's observations, from a scratch directory; returns the evaluation rows"""
    import contextlib
    import io

    from psvo_amd import runner
    d = data or fixture()
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        with open(os.path.join(tmp, "datadict"), "wb") as f:      # the loader's format: src/utils/data_loader.py:5-45
            pickle.dump({"Ytrain": d["Ytrain"].astype(np.float64), "Yvalid": d["Yvalid"].astype(np.float64)}, f)
        FLAGS = notebook_flags(seed, epochs, datadir=tmp + "/", datadict="datadict", **flag_overrides)
        os.chdir(tmp)
        try:
            t0 = time.time()
            sink = io.StringIO()
            with (contextlib.redirect_stdout(sink) if quiet else contextlib.nullcontext()):
                hist = runner.main(FLAGS)
            wall = time.time() - t0
        finally:
            os.chdir(cwd)
    rows = _history_rows(hist, FLAGS.print_freq)
    rows.update(seed=seed, epochs=epochs, wall_s=wall, backend="hip")
    return rows


# ------------------------------------------------------------------------------------------------------------------
class _OracleAdam:
    """tf.train.AdamOptimizer(lr) (src/trainer.py:117): lr_t = lr sqrt(1 - b2^t) / (1 - b1^t);
    theta -= lr_t m / (sqrt(v) + eps)  -- TF's epsilon-hat placement; ascent on log_ZSMC (the reference minimises its negative)"""

    def __init__(self, leaves, b1=0.9, b2=0.999, eps=1e-8):
        self.leaves, self.b1, self.b2, self.eps, self.t = leaves, b1, b2, eps, 0
        self.m = [torch.zeros_like(p) for p in leaves]
        self.v = [torch.zeros_like(p) for p in leaves]

    def step(self, lr):
        self.t += 1
        lr_t = lr * math.sqrt(1 - self.b2 ** self.t) / (1 - self.b1 ** self.t)
        with torch.no_grad():
            for p, m, v in zip(self.leaves, self.m, self.v):
                if p.grad is None:                           # (no path to the objective: TF leaves such a variable alone)
                    continue
                g = -p.grad                                  # gradient of the loss -log_ZSMC
                m.mul_(self.b1).add_(g, alpha=1 - self.b1)
                v.mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
                p.sub_(lr_t * m / (v.sqrt() + self.eps))
                p.grad = None


def _leaves(x, out):
    if torch.is_tensor(x):
        out.append(x)
    elif isinstance(x, dict):
        for v in x.values():
            _leaves(v, out)
    elif isinstance(x, (list, tuple)):
        for v in x:
            _leaves(v, out)
    return out


def run_oracle(seed, epochs, data=None, dtype=torch.float64, progress=None, max_steps=None):
    """the reference's training loop (src/trainer.py:100-195) with the CPU oracle as the compute"""
    from oracle import psvo_oracle as O
    from psvo_amd.model import SSM
    from tests import helpers as Hh
    d = data or fixture()
    FLAGS = notebook_flags(seed, epochs)
    torch.manual_seed(FLAGS.seed)                 # psvo_amd/runner.py: seeds, then the SSM is initialised on the host
    np.random.seed(FLAGS.seed)
    P = SSM(FLAGS).export_reference_layout(dtype)
    seen, leaves = set(), []
    for p in _leaves(P, []):                      # (f == q1 under use_bootstrap: shared tensors appear once)
        if p.is_floating_point() and id(p) not in seen:
            seen.add(id(p))
            leaves.append(p.requires_grad_(True))
    fl = Hh.oracle_flags(FLAGS, "PSVO")
    smc = O.OraclePSVO(P, fl)
    opt = _OracleAdam(leaves)
    gen = torch.Generator().manual_seed(FLAGS.seed)
    obs_train = torch.tensor(d["Ytrain"]).to(dtype)
    obs_valid = torch.tensor(d["Yvalid"]).to(dtype)
    T = obs_train.shape[1]
    hist = {"log_ZSMC_trains": [], "log_ZSMC_tests": [], "R_square_trains": [], "R_square_tests": []}
    sched = {"best": 0, "early": 0, "reduce": 0, "lr": FLAGS.lr}

    def noise(B):
        return O.make_noise(fl, B, T, seed=int(torch.randint(0, 2 ** 31 - 1, (1,), generator=gen)), dtype=dtype)

    def evaluate():
        with torch.no_grad():
            for key, obs in (("train", obs_train), ("test", obs_valid)):
                z, log = smc.get_log_ZSMC(obs, noise(obs.shape[0]))
                y_hat, y = smc.n_step_prediction(FLAGS.MSE_steps, log["Xs"], obs)
                hist["log_ZSMC_%ss" % key].append(float(z))
                hist["R_square_%ss" % key].append(O.evaluate_R_square(y_hat, y).numpy())

    def adjust_lr():                               # src/trainer.py:244-270
        h = hist["log_ZSMC_tests"]
        best, latest = int(np.argmax(h)), len(h) - 1
        if best != sched["best"]:
            sched.update(best=best, early=0, reduce=0)
        if best != latest:
            sched["early"] += 1
            sched["reduce"] += 1
            if sched["early"] * FLAGS.print_freq == FLAGS.early_stop_patience:
                return False
            if sched["reduce"] * FLAGS.print_freq == FLAGS.lr_reduce_patience:
                sched["reduce"] = 0
                sched["lr"] = max(sched["lr"] * FLAGS.lr_reduce_factor, FLAGS.min_lr)
        return True

    t0 = time.time()
    steps = 0
    order = np.arange(len(obs_train))
    for i in range(epochs):
        if i == 0:
            evaluate()
        order = order[np.random.permutation(len(order))]      # (sklearn.utils.shuffle re-shuffles the shuffled arrays)
        for j in order:
            z, _ = smc.get_log_ZSMC(obs_train[j:j + 1], noise(1))
            z.backward()
            opt.step(sched["lr"])
            steps += 1
            if max_steps is not None and steps >= max_steps:
                break
        if max_steps is not None and steps >= max_steps:
            break
        if (i + 1) % FLAGS.print_freq == 0:
            evaluate()
            go = adjust_lr()
            if progress:
                progress(i + 1, hist, time.time() - t0)
            if not go:
                break
        elif progress:
            progress(i + 1, None, time.time() - t0)
    rows = _history_rows(hist, FLAGS.print_freq)
    rows.update(seed=seed, epochs=epochs, wall_s=time.time() - t0, backend="oracle-fp64" if dtype == torch.float64 else "oracle-fp32",
                steps=steps)
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", choices=("hip", "oracle"), required=True)
    ap.add_argument("--seeds", type=int, nargs="+", default=[0])
    ap.add_argument("--epochs", type=int, default=10)
    ap.add_argument("--threads", type=int, default=1)
    ap.add_argument("--out", required=True)
    ap.add_argument("--dtype", choices=("fp64", "fp32"), default="fp64", help="oracle backend: arithmetic type")
    ap.add_argument("--verbose", action="store_true", help="hip backend: let the runner print its per-epoch lines")
    a = ap.parse_args()
    d = fixture()
    out = {"config": {k: v for k, v in NOTEBOOK_FLAGS.items()}, "notebook": notebook_rows(d, upto=max(a.epochs, 10)),
           "runs": []}

    def dump():
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out, "w") as f:
            json.dump(out, f, indent=1)

    for seed in a.seeds:
        if a.backend == "hip":
            rows = run_hip(seed, a.epochs, data=d, quiet=not a.verbose)
        else:
            torch.set_num_threads(a.threads)

            def progress(epoch, hist, wall):
                print("oracle seed %d epoch %d  %.0f s%s" % (seed, epoch, wall, "" if hist is None else
                      "  valid log_ZSMC %.3f" % hist["log_ZSMC_tests"][-1]), flush=True)
            rows = run_oracle(seed, a.epochs, data=d, progress=progress,
                              dtype=torch.float64 if a.dtype == "fp64" else torch.float32)
        out["runs"].append(rows)
        print(json.dumps({k: rows[k] for k in ("backend", "seed", "wall_s", "iter", "valid_log_ZSMC", "valid_Rsq_k0")}), flush=True)
        dump()
    dump()


if __name__ == "__main__":
    main()
