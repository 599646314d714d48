"""The reference's one published outcome for the path -- the training log of notebooks/PSVO.ipynb -- as a test.

tests/golden/fhn_notebook.npz (tests/golden/make_fhn_slice.py) carries the 200 + 40 observation sequences of the reference's
default data set and the 40 evaluation lines the notebook printed (PSVO, N = 16, M = 8, batch 1, T = 200, lr 3e-3, evaluation
every 10 epochs): iter 1 -778.3 / -775.1, iter 10 -380.8 / **-380.5**, then a plateau between -380.5 and -399.5 until iter 130,
best -361.1 at iter 190, R^2(k = 0) 0.92 at iter 10 and 0.984 at iter 190.

Two tests.
  * `test_training_tracks_the_oracle_step_by_step`: the whole training step -- forward filter, backward simulation, objective,
    hand-written reverse pass, Adam -- against the fp64 oracle + torch autograd + the same Adam, from the same initial
    parameters, on the same sequences, with IDENTICAL injected noise (the oracle's draws teacher-forced into the kernels), step
    after step: the per-step ELBOs and the parameter vectors must stay together.  Deterministic; this is the parity statement.
  * `test_notebook_curve_at_iter_10`: `psvo_amd.runner.main` end to end with the notebook's flags for 10 epochs (2000 training
    steps, 6.7 s on one MI355X; the notebook's TF-CPU run took ~1100 s) over SEEDS; statistical (TF's initial weights and
    draws cannot be replayed): every run that trained must land inside BAND of the notebook's -380.5 and reach its R^2(k = 0)
    of 0.92 to within 0.05.

About the bands.  One evaluation of one trained model is a noisy number: the notebook's own consecutive evaluations of an
(almost) converged model scatter over -380.5 ... -399.5 between iter 10 and iter 130 (same neighbourhood of parameters, fresh
draws each time), and where a run stands after 10 epochs depends on its initialisation: eight HIP seeds recorded in
profiles/r03_notebook_curve_hip_10ep.json give -370.9 ... -431.3 at iter 10 (mean -394.6, standard deviation 17.0, median
-393.2); the fp64 ORACLE trained by the same loop on the CPU gives -403.6 and -384.7 (seeds 1, 2), the fp32 oracle -372.4
(seed 3; profiles/r03_notebook_curve_oracle*.json).  The notebook's -380.5 is one draw from that distribution (0.8 sigma from
its mean).  Hence: every trained seed within BAND = 60 (3.5 sigma), the median of the trained seeds within MEDIAN_BAND = 25.
Longer runs (profiles/r03_notebook_curve_hip_400ep.json): seed 1 follows the notebook's whole curve -- -370.9 at iter 10,
-361.4 at iter 130, -354.3 at iter 290 (notebook: -380.5, -384.2, best -361.1 at iter 190), R^2(k = 0) 0.98 -- while seeds 0 and 3
leave the basin between iter 10 and iter 20 (see below).

About runs that do NOT train.  With he_normal kernels the freshly initialised transition MLP has a gain of about sqrt(2)
per step, and for roughly half of all seeds the k = 30 prediction of the initial model already explodes (R^2(k = 30) of
-8e2 ... -7e13 at iter 1 here; the notebook's seed happened to be tame: -0.14).  From such a start the forward filter's particles can
leave the data range by orders of magnitude (|X| ~ 1e3, log-weights ~ -7e4), the objective's gradient becomes a difference of
huge nearly equal terms, and fp32 -- the reference's own arithmetic type -- returns garbage for it: on the parameter state recorded
one step before such an event (tests/golden/fhn_illconditioned_state.npz) the fp32 ORACLE is off by 1.07e5 on q1's output
kernel (true scale 1.2e2) and the HIP path by 1.03e5, tensor by tensor alike (`test_illconditioned_state_is_fp32s_own_error`).
Such a run can end with an evaluation ELBO of -1e30 or stop on a non-finite one.  That is a property of the algorithm in fp32
at this learning rate, not of this implementation -- the fp32 ORACLE trained by the same loop on the CPU shows it too (seed 5:
valid log_ZSMC -515 and R^2(k = 0) -0.34 at iter 10, profiles/r03_notebook_curve_oracle32_seed5.json); the test therefore
requires MIN_TRAINED of the seeds to train and holds every trained one to the band.
"""
import os

import numpy as np
import pytest
import torch

from oracle import psvo_oracle as O
from tests import helpers as Hh
from tests import notebook_curve as NC
from tests import test_gpu_parity as TP

pytestmark = pytest.mark.gpu

SEEDS = (0, 1, 2, 3, 4, 5)
MIN_TRAINED = 4
BAND = 60.0          # every trained seed (3.5 sigma of the seed-to-seed spread, see the module docstring)
MEDIAN_BAND = 25.0   # the median over the trained seeds
GOLD_STATE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fhn_illconditioned_state.npz")


def _trained(valid_elbo):
    return np.isfinite(valid_elbo) and valid_elbo > -600.0      # (a fresh model sits at -640 ... -850)


def test_notebook_curve_at_iter_10(built_lib):
    d = NC.fixture()
    nb = dict(zip(d["nb_iter"].tolist(), zip(d["nb_valid_log_ZSMC"].tolist(), d["nb_valid_Rsq"][:, 0].tolist())))
    assert nb[10] == (-380.498, 0.92255563)
    runs = [NC.run_hip(seed, 10, data=d) for seed in SEEDS]
    for r in runs:
        # (a run whose training ELBO is not finite at the evaluation stops there, like the reference's trainer -- "Nan in
        #  log_ZSMC, stop training", src/trainer.py:221-223 -- and has no second row)
        assert r["iter"] in ([1, 10], [1]), r["iter"]
    # a freshly initialised model: the notebook's -775.1 must lie inside the range of the fresh-init values (sanity)
    fresh = [r["valid_log_ZSMC"][0] for r in runs]
    assert min(fresh) < d["nb_valid_log_ZSMC"][0] < max(fresh), fresh
    nan = float("nan")
    at10 = {r["seed"]: ((r["valid_log_ZSMC"][1], r["valid_Rsq_k0"][1], r["train_log_ZSMC"][1]) if len(r["iter"]) == 2
                        else (nan, nan, nan)) for r in runs}
    trained = {s: v for s, v in at10.items() if _trained(v[0])}
    assert len(trained) >= MIN_TRAINED, at10
    for s, (elbo, r2, elbo_train) in trained.items():
        assert abs(elbo - nb[10][0]) <= BAND, (s, elbo, nb[10][0])
        assert abs(elbo_train - float(d["nb_train_log_ZSMC"][1])) <= BAND, (s, elbo_train)
        assert r2 >= nb[10][1] - 0.05, (s, r2, nb[10][1])
    # the median over the trained seeds is a sharper statement than any single run
    med = float(np.median([v[0] for v in trained.values()]))
    assert abs(med - nb[10][0]) <= MEDIAN_BAND, (med, at10)


def _hip_trainer(FLAGS, model):
    from psvo_amd.optim import FlatParams, TFAdam
    from psvo_amd.SMC.PSVO import PSVO
    smc = PSVO(model, FLAGS)
    flat = FlatParams(model)
    return smc, flat, TFAdam(flat)


def test_training_tracks_the_oracle_step_by_step(built_lib):
    from psvo_amd.model import SSM
    STEPS = 24
    d = NC.fixture()
    FLAGS = NC.notebook_flags(seed=1, epochs=1)
    torch.manual_seed(FLAGS.seed)
    model = SSM(FLAGS)
    P = model.export_reference_layout(torch.float64)
    seen, leaves = set(), []
    for p in NC._leaves(P, []):
        if p.is_floating_point() and id(p) not in seen:
            seen.add(id(p))
            leaves.append(p.requires_grad_(True))
    fl = Hh.oracle_flags(FLAGS, "PSVO")
    oracle = O.OraclePSVO(P, fl)
    o_opt = NC._OracleAdam(leaves)
    model = model.cuda()
    smc, flat, h_opt = _hip_trainer(FLAGS, model)
    order = np.random.RandomState(0).permutation(200)[:STEPS]
    worst_rel, worst_par = 0.0, 0.0
    for k, j in enumerate(order):
        obs = torch.tensor(d["Ytrain"][j:j + 1]).double()
        noise = O.make_noise(fl, 1, 200, seed=5000 + k)
        z_o, log_o = oracle.get_log_ZSMC(obs, noise)
        z_o.backward()
        teacher = {"idx_f": log_o["idx_f"], "idx_b": log_o["idx_b"]}
        nz = Hh.noise_to_hip({**noise, **teacher}, "cuda")
        for key in ("u_f", "u_b"):
            nz.pop(key, None)
        flat.zero_grad()
        z_h, _ = smc.get_log_ZSMC(obs.float().cuda(), None, noise=nz)
        z_h.backward()
        # gradients of THIS step before either optimizer moves
        g_o = {n: (torch.zeros_like(ref) if ref.grad is None else ref.grad.clone()) for n, _, ref in TP._pairs(model, P)}
        g32 = None
        for n, p, ref in TP._pairs(model, P):
            g = torch.zeros_like(ref) if p.grad is None else p.grad.detach().double().cpu()
            scale = max(float(g_o[n].abs().max()), 1e-6)
            err = float((g - g_o[n]).abs().max())
            if err > 2e-3 * scale + 1e-6:
                # the measured yardstick of tests/test_gpu_instantiations.py: what the fp32 ORACLE loses on this very step
                # (same parameters -- taken from the HIP replica, which has not moved yet --, noise and indices)
                if g32 is None:
                    _, P32 = TP._oracle_grads(model, FLAGS, "PSVO", obs, noise, teacher, dtype=torch.float32)
                    g32 = {nn: (torch.zeros_like(r) if r32.grad is None else r32.grad.double())
                           for (nn, _, r), (_, _, r32) in zip(TP._pairs(model, P), TP._pairs(model, P32))}
                err32 = float((g32[n] - g_o[n]).abs().max())
                assert err <= 4.0 * err32, (k, n, err, scale, err32)
        o_opt.step(FLAGS.lr)
        h_opt.step(FLAGS.lr)
        torch.cuda.synchronize()
        rel = abs(float(z_h.detach()) - float(z_o)) / abs(float(z_o))
        worst_rel = max(worst_rel, rel)
        assert rel <= 1e-4, (k, float(z_h.detach()), float(z_o))
        for n, p, ref in TP._pairs(model, P):
            worst_par = max(worst_par, float((p.detach().double().cpu() - ref.detach()).abs().max()))
    # 24 Adam steps of 3e-3 move a parameter by up to 0.07.  Adam divides by sqrt(v): for a coordinate whose gradient is
    # itself of the size of its fp32 rounding error the update direction is decided by that error, so the replicas may part by
    # a fraction of ONE step there (measured: 1.4e-3 = 0.5 lr on the worst coordinate) -- never by more than one step
    assert worst_par <= FLAGS.lr, worst_par
    assert float(z_o) > -700.0                # and the model has started to learn (fresh: -765)


def test_illconditioned_state_is_fp32s_own_error(built_lib):
    """The parameter state one training step before a 3e6 gradient norm (seed 3, step 425 of a run recorded in round 3): the
    filter's particles have left the data range (|X| up to 1.3e3, log-weights down to -6.9e4).  HIP and the fp32 oracle must
    be wrong TOGETHER against the fp64 oracle -- tensor by tensor within a factor 4 of each other -- while the ELBO still
    agrees to 1e-4."""
    from psvo_amd.model import SSM
    from psvo_amd.optim import FlatParams
    from psvo_amd.SMC.PSVO import PSVO
    st = np.load(GOLD_STATE)
    FLAGS = NC.notebook_flags(seed=3, epochs=1)
    torch.manual_seed(3)
    model = SSM(FLAGS).cuda()
    fp = FlatParams(model)
    assert fp.numel == st["flat"].size
    fp.flat.copy_(torch.tensor(st["flat"]).cuda())
    obs = torch.tensor(st["obs"][None]).double()
    fl = Hh.oracle_flags(FLAGS, "PSVO")
    noise = O.make_noise(fl, 1, 200, seed=int(st["noise_seed"]))
    _, ref0 = Hh.run_oracle(model, FLAGS, "PSVO", obs, noise)
    assert float(ref0["X_prevs"].abs().max()) > 1e3 and float(ref0["log_Ws"].min()) < -6e4
    teacher = {"idx_f": ref0["idx_f"], "idx_b": ref0["idx_b"]}
    z64, P64 = TP._oracle_grads(model, FLAGS, "PSVO", obs, noise, teacher)
    z32, P32 = TP._oracle_grads(model, FLAGS, "PSVO", obs, noise, teacher, dtype=torch.float32)
    nz = Hh.noise_to_hip({**noise, **teacher}, "cuda")
    for key in ("u_f", "u_b"):
        nz.pop(key, None)
    fp.zero_grad()
    z, _ = PSVO(model, FLAGS).get_log_ZSMC(obs.float().cuda(), None, noise=nz)
    z.backward()
    torch.cuda.synchronize()
    assert abs(float(z.detach()) - float(z64)) <= 1e-4 * abs(float(z64))
    big = 0
    for (name, p, ref), (_, _, r32) in zip(TP._pairs(model, P64), TP._pairs(model, P32)):
        if ref.grad is None:
            continue
        g = p.grad.detach().double().cpu()
        scale = float(ref.grad.abs().max())
        e_hip, e_32 = float((g - ref.grad).abs().max()), float((r32.grad.double() - ref.grad).abs().max())
        assert e_hip <= max(2e-3 * scale + 1e-6, 4.0 * e_32), (name, e_hip, e_32, scale)
        big += e_32 > 10.0 * scale
    assert big >= 5        # the state IS ill-conditioned: fp32 itself is off by > 10x the true gradient on several tensors
