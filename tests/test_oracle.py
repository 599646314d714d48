"""CPU tests of the oracle itself: known-answer tests with analytic answers, a finite-difference
check of its autograd, and the committed golden fixtures (regression pin).  The reference has no
tests or fixtures for this path (parity unpinned at the TFP boundary), so these KATs are what pins
the restatement's arithmetic."""
import math
import os
import sys

import numpy as np
import pytest
import torch

from oracle import psvo_oracle as O

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import make_golden as MG  # noqa: E402


def _flags(obj, **kw):
    fl = dict(Dx=2, Dy=1, n_particles=8, n_particles_for_BSim_proposal=4, use_bootstrap=True, use_2_q=True,
              objective=obj, layers=[8], y_smoother_Dhs=[4], X0_smoother_Dhs=[4])
    fl.update(kw)
    return fl


def test_logsumexp_and_multinomial_unit_vectors():
    logW = torch.log(torch.tensor([[0.1], [0.2], [0.3], [0.4]], dtype=torch.float64))
    assert abs(float(O.logsumexp(logW, 0)) - 0.0) < 1e-12
    # cdf = [.1, .3, .6, 1.0] (times a common factor): idx = #{cdf <= u}
    u = torch.tensor([[0.05], [0.11], [0.29], [0.31], [0.59], [0.61], [0.999]], dtype=torch.float64)
    idx = O.multinomial_idx(logW, u)
    assert idx.flatten().tolist() == [0, 1, 1, 2, 2, 3, 3]
    # sub-particle form: classes on axis 0, one draw per batch element
    lw = torch.log(torch.tensor([[0.5, 0.1], [0.5, 0.9]], dtype=torch.float64))
    assert O.multinomial_idx(lw, torch.tensor([0.49, 0.2], dtype=torch.float64)).tolist() == [0, 1]


def test_diag_log_prob_matches_scipy():
    from scipy.stats import multivariate_normal
    g = torch.Generator().manual_seed(0)
    x = torch.randn(5, 3, generator=g, dtype=torch.float64)
    mu = torch.randn(5, 3, generator=g, dtype=torch.float64)
    s = torch.rand(3, generator=g, dtype=torch.float64) + 0.5
    lp = O.diag_log_prob(x, mu, s)
    for i in range(5):
        ref = multivariate_normal(mu[i].numpy(), np.diag(s.numpy() ** 2)).logpdf(x[i].numpy())
        assert abs(float(lp[i]) - ref) < 1e-10


def test_bootstrap_without_2q_weights_are_emission_only():
    """use_bootstrap and not use_2_q: f_t - q_t == 0 so log_W_t = g_t - log N (t >= 1) exactly
    (SURVEY.md section 8c KAT)."""
    fl = _flags("AESMC", use_2_q=False)
    P = O.make_params(fl, seed=1, bias_scale=0.3)
    _, obs = O.fhn_synthetic(2, 6, seed=0)
    noise = O.make_noise(fl, 2, 6, seed=2)
    o = O.OracleAESMC(P, fl)
    _, log = o.get_log_ZSMC(obs, noise)
    for t in range(1, 6):
        g = o.g.log_prob(log["X_prevs"][t], obs[:, t])
        assert torch.allclose(log["log_Ws"][t], g - math.log(8.0), atol=1e-12)


def test_iwae_particle_permutation_invariance():
    fl = _flags("IWAE")
    P = O.make_params(fl, seed=3, bias_scale=0.3)
    _, obs = O.fhn_synthetic(2, 5, seed=1)
    noise = O.make_noise(fl, 2, 5, seed=4)
    z1, _ = O.OracleIWAE(P, fl).get_log_ZSMC(obs, noise)
    perm = torch.randperm(8, generator=torch.Generator().manual_seed(0))
    noise2 = {"eps_f": noise["eps_f"][:, perm], "u_f": noise["u_f"][:, perm]}
    z2, _ = O.OracleIWAE(P, fl).get_log_ZSMC(obs, noise2)
    assert abs(float(z1) - float(z2)) < 1e-10


def test_single_subparticle_gives_omega_equal_q():
    """M = 1: normalised omega == 0 and Omega == q (+ log 1)."""
    fl = _flags("PSVO", n_particles_for_BSim_proposal=1)
    P = O.make_params(fl, seed=5, bias_scale=0.3)
    _, obs = O.fhn_synthetic(2, 5, seed=2)
    noise = O.make_noise(fl, 2, 5, seed=6)
    o = O.OraclePSVO(P, fl)
    _, log = o.get_log_ZSMC(obs, noise)
    # recompute q of the single proposal at t = T-1 and compare with Omega
    _, enc = o.BS_preprocess_obs(obs)
    x, q = o.BSim_q_init.sample_and_log_prob(enc[-1], noise["eps_b"][4])
    assert torch.allclose(log["bw_log_Omegas"][4], q[0], atol=1e-12)
    assert torch.allclose(log["bw_Xs"][4], x[0], atol=1e-12)


def _kalman_loglik(A, C, sf2, sg2, m0, P0, ys):
    m, Pm, ll = m0, P0, 0.0
    for t, y in enumerate(ys):
        if t > 0:
            m, Pm = A @ m, A @ Pm @ A.T + sf2
        S = C @ Pm @ C.T + sg2
        r = y - C @ m
        ll += -0.5 * (r @ np.linalg.solve(S, r) + np.log(np.linalg.det(2 * np.pi * S)))
        K = Pm @ C.T @ np.linalg.inv(S)
        m, Pm = m + K @ r, (np.eye(len(m)) - K @ C) @ Pm
    return ll


def test_linear_gaussian_ssm_matches_kalman_likelihood():
    """With zero hidden layers the MLPs are linear: bootstrap / no-2q AESMC is a bootstrap particle
    filter of a linear-Gaussian SSM whose exact log-likelihood the Kalman filter gives; the SMC
    estimate converges to it as N grows (SURVEY.md section 8c KAT)."""
    Dx, Dy, T, B, N = 2, 1, 6, 2, 6000
    fl = _flags("AESMC", Dx=Dx, Dy=Dy, n_particles=N, use_2_q=False, layers=[])
    P = O.make_params(fl, seed=9, bias_scale=0.0)
    g = torch.Generator().manual_seed(1)
    A = torch.tensor([[0.9, 0.2], [-0.1, 0.8]], dtype=torch.float64)
    C = torch.tensor([[1.0], [0.5]], dtype=torch.float64)            # (Dx, Dy) keras kernel of g
    P["q1"]["mu"] = (A.t().contiguous(), torch.zeros(2, dtype=torch.float64))   # x W = A x
    P["g"]["mu"] = (C, torch.zeros(1, dtype=torch.float64))
    P["q0"]["mu"] = (A.t().contiguous(), torch.zeros(2, dtype=torch.float64))   # proposal == prior at t = 0
    for k in ("q0", "q1", "g"):
        P[k]["sigma_raw"] = torch.full_like(P[k]["sigma_raw"], 0.3)
        P[k]["sigma_min"] = 0.7
    sig = max(math.log1p(math.exp(0.3)), 0.7)                         # max(softplus(raw), sigma_min)
    obs = torch.randn(B, T, Dy, generator=g, dtype=torch.float64)
    noise = O.make_noise(fl, B, T, seed=5)
    z, _ = O.OracleAESMC(P, fl).get_log_ZSMC(obs, noise)
    W0, b0 = P["X0_transformer"]
    ll = 0.0
    for b in range(B):
        z0 = (obs[b, 0] @ W0 + b0).numpy()
        ll += _kalman_loglik(A.numpy(), C.t().numpy(), sig ** 2 * np.eye(2), sig ** 2 * np.eye(1),
                             A.numpy() @ z0, sig ** 2 * np.eye(2), obs[b].numpy())
    assert abs(float(z) - ll / B) < 0.05, (float(z), ll / B)


@pytest.mark.parametrize("objective,poisson", [("PSVO", False), ("PSVOwR", False), ("PSVO", True)])
def test_oracle_autograd_matches_finite_differences(objective, poisson):
    """teacher-forced indices make log_ZSMC a smooth function of the parameters (also with the tf_poisson emission)"""
    fl = _flags(objective, n_particles=5, n_particles_for_BSim_proposal=3, poisson_emission=poisson)
    Oracle = O.OBJECTIVES[objective]
    P = O.make_params(fl, seed=2, bias_scale=0.3)
    for k in P:
        if isinstance(P[k], dict) and "sigma_raw" in P[k]:
            P[k]["sigma_raw"] = torch.full_like(P[k]["sigma_raw"], 1.0)
            P[k]["sigma_min"] = 0.2
    _, obs = O.fhn_synthetic(2, 4, seed=3)
    noise = O.make_noise(fl, 2, 4, seed=8)
    with torch.no_grad():
        _, log = Oracle(P, fl).get_log_ZSMC(obs, noise)
    teacher = {**noise, "idx_f": log["idx_f"], "idx_b": log["idx_b"]}
    if objective == "PSVOwR":
        teacher["idx_r"] = log["idx_r"]
    targets = [P["q1"]["layers"][0][0], P["g"]["mu"][1], P["q1_inv"]["sigma_raw"], P["BSim_q2"]["mu"][0],
               P["bRNN"]["y_smoother"][0]["fw"][0]]
    for t in targets:
        t.requires_grad_(True)
    z, _ = Oracle(P, fl).get_log_ZSMC(obs, teacher)
    grads = torch.autograd.grad(z, targets)
    for t, g in zip(targets, grads):
        flat = t.detach().view(-1)
        for k in (0, flat.numel() // 2):
            old = float(flat[k])
            with torch.no_grad():
                flat[k] = old + 1e-6
                zp, _ = Oracle(P, fl).get_log_ZSMC(obs, teacher)
                flat[k] = old - 1e-6
                zm, _ = Oracle(P, fl).get_log_ZSMC(obs, teacher)
                flat[k] = old
            fd = (float(zp) - float(zm)) / 2e-6
            assert abs(fd - float(g.view(-1)[k])) < 1e-5 * max(1.0, abs(fd)), (fd, float(g.view(-1)[k]))


@pytest.mark.parametrize("name", sorted(MG.CASES))
def test_golden_fixture_regression(name):
    """the committed vectors are reproduced bit-for-bit-ish by the current oracle"""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz")
    ref = np.load(path)
    fl, P, obs, noise = MG.build(name)
    z, log = MG.run(fl, P, obs, noise)
    assert abs(float(z) - float(ref["log_ZSMC"])) < 1e-10
    for k in ("Xs", "log_Ws", "bw_Xs", "bw_log_Omegas", "bw_X_ancestors", "bw_log_W"):
        if "out." + k in ref.files:
            assert np.allclose(log[k].detach().numpy(), ref["out." + k], atol=1e-10)
    for k in ("idx_f", "idx_b", "idx_r"):
        if "out." + k in ref.files:
            assert (log[k].numpy() == ref["out." + k]).all()
    g = P["q1"]["layers"][0][0].grad
    assert np.allclose(g.numpy(), ref["grad.q1.layers.0.0"], atol=1e-9)


def test_psvowr_weight_closed_form():
    """The HIP kernel writes bw_log_W = logsumexp_m(omega_raw) - phi_sel - log M (csrc/psvowr_fwd.hip); the oracle
    restates the reference's expression (PSVOwR.py:135-142) term by term.  With the cross-chain draw teacher-forced
    to the identity and M = 1 the weights collapse to Lambda + g - q of the single proposal draw, so that
    log_ZSMC(M = 1) is an importance-sampling estimate whose value is independent of the sub-particle index."""
    fl = _flags("PSVOwR", n_particles=6, n_particles_for_BSim_proposal=1)
    P = O.make_params(fl, seed=4, bias_scale=0.2)
    _, obs = O.fhn_synthetic(2, 5, seed=1)
    noise = O.make_noise(fl, 2, 5, seed=9)
    with torch.no_grad():
        z, log = O.OraclePSVOwR(P, fl).get_log_ZSMC(obs, noise)
    assert (log["idx_b"] == 0).all()                 # one sub-particle: nothing to choose
    assert torch.isfinite(z)
    # the cross-chain ancestors really index the selected sub-particles of the same step
    idx = log["idx_r"].unsqueeze(-1).expand(-1, -1, -1, log["bw_Xs"].shape[-1])
    assert torch.equal(log["bw_X_ancestors"], torch.gather(log["bw_Xs"], 1, idx))


# ---------------------------------------------------------------------------------------------
# output_cov and diag_cov: state-dependent diagonal scales (src/transformation/MLP.py:40-46,58-61, src/distribution/mvn.py:66-71)
# ---------------------------------------------------------------------------------------------
def test_state_dependent_scale_closed_form_and_scipy():
    """scale = sigma_con + 0.1 (exp(h W_s + b_s) + 1e-6): log_prob against scipy at two inputs with different scales"""
    from scipy.stats import norm
    gen = torch.Generator().manual_seed(5)
    p = O.make_mlp(gen, 2, [8], 3, 1.0, 0.2, torch.float64, bias_scale=0.3, cov_head=True)
    d = O.OracleMVN(p)
    x = torch.randn(4, 2, generator=gen, dtype=torch.float64)
    y = torch.randn(4, 3, generator=gen, dtype=torch.float64)
    mu, s = d.mean_and_sigma(x)
    h = torch.relu(x @ p["layers"][0][0] + p["layers"][0][1])
    want = torch.nn.functional.softplus(p["sigma_raw"]).clamp(min=0.2) + 0.1 * (torch.exp(h @ p["sigma"][0] + p["sigma"][1]) + 1e-6)
    assert torch.allclose(s, want) and s.shape == (4, 3) and float((s[0] - s[1]).abs().max()) > 1e-3
    lp = d.log_prob(x, y)
    ref = norm.logpdf(y.numpy(), mu.numpy(), s.numpy()).sum(-1)
    assert np.allclose(lp.numpy(), ref, atol=1e-12)
    # sample: mean + scale * eps, and its density
    eps = torch.randn(4, 3, generator=gen, dtype=torch.float64)
    smp, q = d.sample_and_log_prob(x, eps)
    assert torch.allclose(smp, mu + s * eps) and np.allclose(q.numpy(), norm.logpdf(eps.numpy()).sum(-1) - np.log(s.numpy()).sum(-1))


def test_constant_cov_head_equals_a_larger_state_independent_scale():
    """a covariance head with a zero kernel adds the constant 0.1 (e^b + 1e-6) to every scale: the model must then equal the
    plain model whose sigma_con is larger by exactly that -- for every objective, values and ELBO"""
    for objective in ("AESMC", "PSVO", "PSVOwR"):
        fl = _flags(objective, n_particles=6, n_particles_for_BSim_proposal=3, output_cov=True, diag_cov=True)
        P = O.make_params(fl, seed=4, bias_scale=0.2)
        plain = O.make_params({**fl, "output_cov": False}, seed=4, bias_scale=0.2)
        for k, v in P.items():
            if isinstance(v, dict) and "sigma" in v:
                W, b = v["sigma"]
                v["sigma"] = (torch.zeros_like(W), b)
                add = 0.1 * (torch.exp(b) + 1e-6)
                # softplus(raw) = 5.0067 > sigma_min: shift the clipped scale through sigma_min (max() then picks it)
                plain[k]["sigma_min"] = None
                plain[k]["sigma_raw"] = None
                plain[k]["_scale"] = torch.nn.functional.softplus(v["sigma_raw"]).clamp(min=v["sigma_min"]) + add
        # (the two parameter sets were drawn from different streams once the heads exist: copy the shared tensors over)
        def copy(dst, src):
            for k, v in src.items():
                if k == "sigma":
                    continue
                if isinstance(v, dict):
                    copy(dst[k], v)
                elif k not in ("sigma_raw", "sigma_min"):
                    dst[k] = v
        copy(plain, P)
        real = O.get_sigma
        try:
            O.get_sigma = lambda p: p["_scale"] if p.get("_scale") is not None else real(p)
            _, obs = O.fhn_synthetic(2, 5, seed=1)
            noise = O.make_noise(fl, 2, 5, seed=3)
            with torch.no_grad():
                z1, l1 = O.OBJECTIVES[objective](P, fl).get_log_ZSMC(obs, noise)
                z0, l0 = O.OBJECTIVES[objective](plain, {**fl, "output_cov": False}).get_log_ZSMC(obs, noise)
        finally:
            O.get_sigma = real
        assert abs(float(z1) - float(z0)) < 1e-10 * abs(float(z0)), (objective, float(z1), float(z0))
        assert torch.allclose(l1["Xs"], l0["Xs"], atol=1e-12)


@pytest.mark.parametrize("objective", ["AESMC", "SVO", "PSVO"])
def test_state_dependent_scale_autograd_matches_finite_differences(objective):
    fl = _flags(objective, n_particles=5, n_particles_for_BSim_proposal=3, output_cov=True, diag_cov=True)
    Oracle = O.OBJECTIVES[objective]
    P = O.make_params(fl, seed=2, bias_scale=0.3)
    for k in P:
        if isinstance(P[k], dict) and "sigma_raw" in P[k]:
            P[k]["sigma_raw"] = torch.full_like(P[k]["sigma_raw"], 1.0)
            P[k]["sigma_min"] = 0.2
            W, b = P[k]["sigma"]
            P[k]["sigma"] = (0.3 * W, b)          # (he_normal kernels through exp(): keep the scales O(1))
    _, obs = O.fhn_synthetic(2, 4, seed=3)
    noise = O.make_noise(fl, 2, 4, seed=8)
    with torch.no_grad():
        _, log = Oracle(P, fl).get_log_ZSMC(obs, noise)
    teacher = {**noise, "idx_f": log["idx_f"]}
    if objective == "PSVO":
        teacher["idx_b"] = log["idx_b"]
    targets = [P["q1"]["sigma"][0], P["g"]["sigma"][1], P["q1"]["layers"][0][0], P["q2"]["sigma"][0], P["q0"]["sigma_raw"]]
    if objective == "PSVO":
        targets += [P["q1_inv"]["sigma"][0], P["BSim_q2"]["sigma"][1]]
    for t in targets:
        t.requires_grad_(True)
    z, _ = Oracle(P, fl).get_log_ZSMC(obs, teacher)
    grads = torch.autograd.grad(z, targets)
    for t, g in zip(targets, grads):
        flat = t.detach().view(-1)
        for k in (0, flat.numel() // 2):
            old = float(flat[k])
            with torch.no_grad():
                flat[k] = old + 1e-6
                zp, _ = Oracle(P, fl).get_log_ZSMC(obs, teacher)
                flat[k] = old - 1e-6
                zm, _ = Oracle(P, fl).get_log_ZSMC(obs, teacher)
                flat[k] = old
            fd = (float(zp) - float(zm)) / 2e-6
            assert abs(fd - float(g.view(-1)[k])) < 2e-5 * max(1.0, abs(fd)), (objective, fd, float(g.view(-1)[k]))
