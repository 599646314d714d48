"""GPU parity tests: HIP path (through the C ABI) vs the CPU oracle on identical inputs + noise.

Tolerances (fp32 kernels vs fp64 oracle), stated per quantity:
  particles / trajectories   atol 2e-4 (values O(1..10))
  log-weights, log-densities atol 5e-4
  ELBO                       rel  1e-4  (north-star bar is 1e-3)
Teacher-forced runs inject the oracle's ancestor / sub-particle indices, so every downstream
quantity is comparable; free-running runs check that the in-kernel multinomial draw reproduces
the oracle's indices (up to draws whose uniform lies within rounding of a CDF edge).
"""
import pytest
import torch

from oracle import psvo_oracle as O
from tests import helpers as Hh

pytestmark = pytest.mark.gpu

CASES = [
    # objective, B, T, N, M, Dx, Dy, H, bootstrap, two_q
    ("AESMC", 2, 6, 8, 4, 2, 1, 16, True, True),
    ("AESMC", 3, 9, 64, 4, 2, 1, 32, True, True),
    ("AESMC", 2, 7, 100, 4, 3, 2, 32, False, True),
    ("AESMC", 2, 7, 130, 4, 2, 1, 32, True, False),
    ("AESMC", 2, 5, 300, 4, 4, 1, 64, False, False),
    ("IWAE", 1, 50, 4, 4, 2, 1, 32, True, True),
    ("IWAE", 2, 8, 96, 4, 3, 1, 16, False, True),
    ("SVO", 2, 8, 32, 4, 2, 1, 32, True, True),
    ("SVO", 2, 6, 16, 4, 3, 1, 32, False, False),
    ("PSVO", 2, 6, 8, 4, 2, 1, 16, True, True),
    ("PSVO", 2, 7, 64, 16, 2, 1, 32, True, True),
    ("PSVO", 2, 6, 50, 8, 3, 1, 32, True, True),
    ("PSVO", 1, 5, 36, 32, 4, 2, 32, False, True),
    ("PSVO", 2, 5, 130, 16, 2, 1, 64, True, False),
    ("PSVO", 1, 5, 20, 4, 3, 1, 32, False, False),
    ("PSVOwR", 2, 6, 8, 4, 2, 1, 16, True, True),
    ("PSVOwR", 2, 7, 64, 16, 2, 1, 32, True, True),
    ("PSVOwR", 2, 6, 50, 8, 3, 1, 32, True, True),
    ("PSVOwR", 1, 5, 36, 32, 4, 2, 32, False, True),
    ("PSVOwR", 2, 5, 130, 16, 2, 1, 64, True, False),
    ("PSVOwR", 1, 5, 20, 4, 3, 1, 32, False, False),
]


# encoder wirings other than the default stack_bidirectional_dynamic_rnn (reference src/model.py:168-178,
# src/SMC/SVO.py:342-367, src/SMC/PSVO.py:208-212): (case, extra flags)
ENCODER_CASES = [
    (("SVO", 2, 7, 16, 4, 2, 1, 16, True, True), dict(use_stack_rnn=False, y_smoother_Dhs="8,16", X0_smoother_Dhs="16,8")),
    (("SVO", 2, 6, 12, 4, 3, 1, 32, False, False), dict(use_stack_rnn=False, X0_use_separate_RNN=False)),
    (("PSVO", 2, 6, 16, 8, 2, 1, 32, True, True), dict(BSim_use_single_RNN=True, y_smoother_Dhs="8,16")),
    (("PSVO", 2, 6, 16, 8, 2, 1, 16, True, True), dict(use_stack_rnn=False, y_smoother_Dhs="16,8")),
    (("PSVOwR", 2, 5, 12, 4, 2, 1, 16, True, True), dict(BSim_use_single_RNN=True)),
]


# FLAGS.poisson_emission (reference src/model.py:153-155, src/distribution/poisson.py:27-50): every kernel family,
# both filter kernels (four lanes per particle at N <= 128 / one lane per particle) and the split-hidden bsim kernels
EMISSION_CASES = [
    (("AESMC", 2, 6, 16, 1, 2, 1, 32, True, True), dict(poisson_emission=True)),
    (("SVO", 2, 5, 160, 1, 3, 2, 16, False, True), dict(poisson_emission=True)),
    (("IWAE", 2, 5, 12, 1, 2, 1, 16, True, False), dict(poisson_emission=True)),
    (("PSVO", 2, 6, 16, 8, 2, 1, 32, True, True), dict(poisson_emission=True)),
    (("PSVO", 2, 5, 40, 4, 3, 2, 16, False, False), dict(poisson_emission=True)),
    (("PSVO", 1, 4, 130, 16, 2, 1, 64, True, True), dict(poisson_emission=True)),
    (("PSVOwR", 2, 5, 12, 4, 2, 1, 16, True, True), dict(poisson_emission=True)),
]


# per-particle MLPs of widths the kernels are not instantiated for, and of different widths within one launch: they run at
# the next instantiated width on zero-padded hidden units (SVO._kernel_width); the hoisted MLPs (q0, q2) of odd widths go
# through psvo_dense_*.  Reference: any comma-separated widths parse (src/runner_flag.py:198-206, src/model.py:99-151).
WIDTH_CASES = [
    (("PSVO", 2, 6, 16, 8, 2, 1, 32, False, True), dict(q1_layers="24", f_layers="40", g_layers="16", q0_layers="20", q2_layers="50")),
    (("AESMC", 2, 6, 16, 1, 2, 1, 32, True, True), dict(q1_layers="50", g_layers="12")),
    (("PSVOwR", 2, 5, 12, 4, 2, 1, 16, True, True), dict(q1_layers="10", g_layers="16")),
    (("SVO", 2, 5, 160, 1, 3, 2, 16, False, False), dict(q1_layers="33", f_layers="64", g_layers="7")),
]


# per-particle MLPs with TWO hidden layers (`*_layers="64,64"`, the example of the reference's flag file,
# src/runner_flag.py:50-57; src/transformation/MLP.py:24-38,50-54): psvo_desc.layers = 2 in every kernel family -- both filter
# kernels (four lanes per particle at N <= 128 / H = 32, one lane per particle otherwise), the split-hidden backward
# simulation, the PSVOwR cluster kernels -- equal, unequal and zero-padded widths, and the MFMA weight gradients
# (psvo_mlp2_wgrad).  (q1_layers also shapes q1_inv; f_layers only exists without use_bootstrap.)
DEPTH_CASES = [
    (("AESMC", 2, 6, 16, 1, 2, 1, 32, True, True), dict(q1_layers="32,32", g_layers="32,32")),
    (("SVO", 2, 5, 160, 1, 3, 2, 32, False, False), dict(q1_layers="64,64", f_layers="64,64", g_layers="64,64")),
    (("IWAE", 2, 6, 12, 1, 2, 1, 32, True, False), dict(q1_layers="24,40", g_layers="32,16")),
    (("PSVO", 2, 6, 16, 8, 2, 1, 32, True, True), dict(q1_layers="32,32", g_layers="32,32")),
    (("PSVO", 2, 5, 130, 16, 2, 1, 32, False, True), dict(q1_layers="64,64", f_layers="64,64", g_layers="64,64")),
    (("PSVO", 1, 5, 36, 4, 4, 2, 32, False, False), dict(q1_layers="32,32", f_layers="20,32", g_layers="32,32")),
    (("PSVOwR", 2, 6, 24, 8, 2, 1, 32, True, True), dict(q1_layers="32,32", g_layers="32,32")),
    (("PSVOwR", 1, 5, 36, 4, 3, 1, 32, False, True), dict(q1_layers="64,48", f_layers="64,64", g_layers="40,64")),
    # the register-hungry corners: Dx = 3 / 4 at H = 64, the 512-thread filter kernels (N > 256), the 512-thread PSVOwR build
    (("PSVO", 2, 5, 40, 8, 3, 1, 32, True, True), dict(q1_layers="64,64", g_layers="64,64")),
    (("PSVO", 1, 4, 24, 4, 4, 2, 32, False, True), dict(q1_layers="64,64", f_layers="64,64", g_layers="64,64")),
    (("AESMC", 2, 5, 300, 1, 4, 1, 32, False, True), dict(q1_layers="64,64", f_layers="64,64", g_layers="64,64")),
    (("SVO", 2, 6, 100, 1, 3, 1, 32, True, True), dict(q1_layers="32,32", g_layers="32,32")),
    (("PSVOwR", 2, 5, 130, 16, 2, 1, 32, True, False), dict(q1_layers="32,32", g_layers="32,32")),
    (("PSVOwR", 1, 5, 36, 4, 3, 1, 32, True, True), dict(q1_layers="64,64", g_layers="64,64")),
    # four lanes per particle at H = 64 and at Dx = 4 (shapes the one-layer build leaves to the lane = particle kernels)
    (("AESMC", 2, 6, 64, 1, 2, 1, 32, True, True), dict(q1_layers="64,64", g_layers="64,64")),
    (("SVO", 2, 5, 48, 1, 4, 1, 32, False, True), dict(q1_layers="32,32", f_layers="32,32", g_layers="32,32")),
]


def _setup(obj, B, T, N, M, Dx, Dy, H, bootstrap, two_q, seed=0, **extra):
    from psvo_amd.model import SSM
    from psvo_amd.SMC.SVO import SVO
    from psvo_amd.SMC.PSVO import PSVO
    from psvo_amd.SMC.AESMC import AESMC
    from psvo_amd.SMC.IWAE import IWAE
    from psvo_amd.SMC.PSVOwR import PSVOwR
    cls = {"SVO": SVO, "PSVO": PSVO, "AESMC": AESMC, "IWAE": IWAE, "PSVOwR": PSVOwR}[obj]
    hs = str(H)
    kw = dict(Dx=Dx, Dy=Dy, n_particles=N, n_particles_for_BSim_proposal=M, batch_size=B, time=T,
              q0_layers=hs, q1_layers=hs, q2_layers=hs, f_layers=hs, g_layers=hs,
              y_smoother_Dhs="8", X0_smoother_Dhs="8", use_bootstrap=bootstrap, use_2_q=two_q)
    kw.update(extra)
    FLAGS = Hh.make_flags(obj, **kw)
    torch.manual_seed(seed)
    model = Hh.perturb_(SSM(FLAGS)).cuda()
    smc = cls(model, FLAGS)
    g = torch.Generator().manual_seed(100 + seed)
    obs = torch.randn(B, T, Dy, generator=g, dtype=torch.float64) * 1.5
    noise = O.make_noise(Hh.oracle_flags(FLAGS, obj), B, T, seed=1234 + seed)
    return FLAGS, model, smc, obs, noise


@pytest.mark.parametrize("case", CASES, ids=lambda c: "-".join(map(str, c)))
def test_teacher_forced_parity(built_lib, case):
    obj = case[0]
    FLAGS, model, smc, obs, noise = _setup(*case)
    z_ref, ref = Hh.run_oracle(model, FLAGS, obj, obs, noise)
    teacher = {"idx_f": ref["idx_f"]} if ref["idx_f"] is not None else {}
    if obj in ("PSVO", "PSVOwR"):
        teacher["idx_b"] = ref["idx_b"]
    if obj == "PSVOwR":
        teacher["idx_r"] = ref["idx_r"]
    nz = Hh.noise_to_hip({**noise, **teacher}, "cuda")
    nz.pop("u_f", None) if "idx_f" in nz else None
    nz.pop("u_b", None) if "sel_b" in nz else None
    nz.pop("u_r", None) if "anc_r" in nz else None
    with torch.no_grad():
        z, log = smc.get_log_ZSMC(obs.float().cuda(), None, noise=nz)
    torch.cuda.synchronize()
    filt = log["filter"]
    assert torch.allclose(Hh.part_to_ref(filt["X"]), ref["X_prevs"], atol=2e-4, rtol=1e-5)
    assert torch.allclose(Hh.part_to_ref(filt["Xanc"]), ref["X_ancestors"], atol=2e-4, rtol=1e-5)
    assert torch.allclose(Hh.w_to_ref(filt["logW"]), ref["log_Ws"], atol=5e-4, rtol=1e-5)
    assert torch.allclose(filt["lse"].double().cpu(), torch.logsumexp(ref["log_Ws"], 1), atol=5e-4, rtol=1e-5)
    if obj == "PSVO":
        bs = log["bsim"]
        assert torch.allclose(Hh.part_to_ref(bs["bwX"]), ref["bw_Xs"], atol=2e-4, rtol=1e-5)
        assert torch.allclose(Hh.w_to_ref(bs["flp"]), ref["f_log_probs"], atol=5e-4, rtol=1e-5)
        assert torch.allclose(Hh.w_to_ref(bs["glp"]), ref["g_log_probs"], atol=5e-4, rtol=1e-5)
        assert torch.allclose(Hh.w_to_ref(bs["Omega"]), ref["bw_log_Omegas"], atol=5e-4, rtol=1e-5)
    if obj == "PSVOwR":
        bs = log["bsim"]
        assert torch.allclose(Hh.part_to_ref(bs["bwX"]), ref["bw_Xs"], atol=2e-4, rtol=1e-5)
        assert torch.allclose(Hh.part_to_ref(bs["bwXanc"]), ref["bw_X_ancestors"], atol=2e-4, rtol=1e-5)
        assert torch.allclose(Hh.w_to_ref(bs["bwW"]), ref["bw_log_W"], atol=5e-4, rtol=1e-5)
        assert torch.allclose(bs["lseW"].double().cpu(), torch.logsumexp(ref["bw_log_W"], 1), atol=5e-4, rtol=1e-5)
        assert int(bs["ws"][-1:].view(torch.int32)) == 0, "a cluster barrier of psvo_bsimwr_forward timed out"
    assert torch.allclose(log["Xs"].double().cpu(), ref["Xs"], atol=2e-4, rtol=1e-5)
    assert abs(float(z) - float(z_ref)) <= 1e-4 * abs(float(z_ref))


@pytest.mark.parametrize("case", CASES, ids=lambda c: "-".join(map(str, c)))
def test_free_running_indices(built_lib, case):
    """In-kernel multinomial draws (prefix-sum CDF + search) against the oracle's definition."""
    obj = case[0]
    FLAGS, model, smc, obs, noise = _setup(*case, seed=3)
    z_ref, ref = Hh.run_oracle(model, FLAGS, obj, obs, noise)
    nz = Hh.noise_to_hip(noise, "cuda")
    with torch.no_grad():
        z, log = smc.get_log_ZSMC(obs.float().cuda(), None, noise=nz)
    torch.cuda.synchronize()
    if ref["idx_f"] is not None:
        idx = log["filter"]["idx"].permute(0, 2, 1).cpu().long()
        assert (idx != ref["idx_f"]).float().mean() == 0.0
    if obj in ("PSVO", "PSVOwR"):
        sel = log["bsim"]["sel"].permute(0, 2, 1).cpu().long()
        assert (sel != ref["idx_b"]).float().mean() == 0.0
    if obj == "PSVOwR":
        anc = log["bsim"]["anc"].permute(0, 2, 1).cpu().long()
        assert (anc != ref["idx_r"]).float().mean() == 0.0
        assert int(log["bsim"]["ws"][-1:].view(torch.int32)) == 0, "an exchange poll timed out"
        smc.check_exchange()                      # (the host-side form of the same check)
    assert abs(float(z) - float(z_ref)) <= 1e-4 * abs(float(z_ref))
    assert torch.allclose(log["Xs"].double().cpu(), ref["Xs"], atol=2e-4, rtol=1e-5)


@pytest.mark.parametrize("B,T,Dy,Dhs", [(2, 7, 1, [8]), (3, 40, 2, [32]), (2, 11, 1, [16, 8]), (2, 9, 3, [64, 32])])
def test_bilstm_encoder(built_lib, B, T, Dy, Dhs):
    """persistent bi-LSTM kernel (psvo_bilstm_forward) vs the oracle's LSTMBlockCell restatement"""
    from psvo_amd.model import StackBiRNN
    torch.manual_seed(1)
    enc = Hh.perturb_(StackBiRNN(Dy, Dhs, "y_smoother")).cuda()
    x = torch.randn(B, T, Dy, dtype=torch.float64)
    layers = [{"fw": (f.kernel.detach().double().cpu(), f.bias.detach().double().cpu()),
               "bw": (b.kernel.detach().double().cpu(), b.bias.detach().double().cpu())}
              for f, b in zip(enc.fw, enc.bw)]
    ref = O.stack_bidirectional_rnn(x, layers)
    with torch.no_grad():
        out = enc(x.float().cuda())
    assert torch.allclose(out.double().cpu(), ref, atol=2e-5, rtol=1e-5)


# ---------------------------------------------------------------------------------------------
# gradients: hand-written backward kernels vs torch autograd of the fp64 oracle
# ---------------------------------------------------------------------------------------------
def _oracle_grads(model, FLAGS, obj, obs, noise, teacher, dtype=torch.float64):
    """ELBO and the autograd gradients of the oracle in `dtype` (fp64: the reference values; fp32: the yardstick of what the
    arithmetic type itself costs on a given case -- tests/test_gpu_instantiations.py)"""
    P = model.export_reference_layout(dtype)
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
    o = O.OBJECTIVES[obj](P, Hh.oracle_flags(FLAGS, obj))
    nz = {k: (v.to(dtype) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in {**noise, **teacher}.items()}
    z, _ = o.get_log_ZSMC(obs.to(dtype), nz)
    z.backward()
    return z.detach(), P


def _pairs(model, P):
    """(name, product parameter, oracle tensor) for every trainable variable"""
    out = []

    def dist(name, d):
        tr = d.transformation
        for i, (W, b) in enumerate(zip(tr.kernels, tr.biases)):
            out.append((name + ".W%d" % i, W, P[name]["layers"][i][0]))
            out.append((name + ".b%d" % i, b, P[name]["layers"][i][1]))
        out.append((name + ".Wmu", tr.mu_kernel, P[name]["mu"][0]))
        out.append((name + ".bmu", tr.mu_bias, P[name]["mu"][1]))
        if "sigma" in P[name]:                      # covariance head (output_cov and diag_cov)
            out.append((name + ".Wsigma", tr.sigma_kernel, P[name]["sigma"][0]))
            out.append((name + ".bsigma", tr.sigma_bias, P[name]["sigma"][1]))
        if "sigma_raw" in P[name]:                  # (tf_poisson has no scale variable)
            out.append((name + ".sigma", d.sigma_con, P[name]["sigma_raw"]))
    dist("q0", model.q0_dist); dist("q1", model.q1_dist); dist("g", model.g_dist)
    if model.use_2_q:
        dist("q2", model.q2_dist)
    if not model.use_bootstrap:
        dist("f", model.f_dist)
    if not (model.use_bootstrap and model.use_2_q):
        out.append(("X0_transformer.W", model.X0_transformer_kernel, P["X0_transformer"][0]))
        out.append(("X0_transformer.b", model.X0_transformer_bias, P["X0_transformer"][1]))
    if model.PSVO or model.PSVOwR:
        dist("BSim_q_init", model.Bsim_q_init_dist); dist("q1_inv", model.q1_inv_dist)
        dist("BSim_q2", model.BSim_q2_dist)
    if model.bRNN is not None:
        for nm, s in (("y_smoother", model.y_smoother), ("X0_smoother", model.X0_smoother)):
            if s is None:
                continue
            for i, f in enumerate(s.fw):
                out.append(("%s.fw%d.W" % (nm, i), f.kernel, P["bRNN"][nm][i]["fw"][0]))
                out.append(("%s.fw%d.b" % (nm, i), f.bias, P["bRNN"][nm][i]["fw"][1]))
            for i, b in enumerate(s.bw):          # (none for the forward-only encoder of BSim_use_single_RNN)
                out.append(("%s.bw%d.W" % (nm, i), b.kernel, P["bRNN"][nm][i]["bw"][0]))
                out.append(("%s.bw%d.b" % (nm, i), b.bias, P["bRNN"][nm][i]["bw"][1]))
    return out


def _grad_mismatches(model, P, rtol):
    bad = []
    for name, p, ref in _pairs(model, P):
        g = torch.zeros_like(ref) if p.grad is None else p.grad.detach().double().cpu()
        r = torch.zeros_like(ref) if ref.grad is None else ref.grad
        scale = max(r.abs().max().item(), 1e-6)
        err = (g - r).abs().max().item()
        if err > rtol * scale + 1e-6:
            bad.append((name, err, scale))
    return bad


def _check_grads(model, P, rtol=2e-3, rerun=None):
    bad = _grad_mismatches(model, P, rtol)
    if bad and rerun is not None:   # diagnostic: does a second HIP pass in the same process agree with the oracle?
        rerun()
        again = _grad_mismatches(model, P, rtol)
        assert not bad, "gradient mismatch (name, max abs err, ref scale): %s; second pass: %s" % (bad, again)
    assert not bad, "gradient mismatch (name, max abs err, ref scale): %s" % bad


GRAD_CASES = CASES


@pytest.mark.parametrize("case", GRAD_CASES, ids=lambda c: "-".join(map(str, c)))
def test_filter_gradients(built_lib, case):
    """d log_ZSMC / d(all parameters) for SVO / AESMC / IWAE / PSVO / PSVOwR, teacher-forced indices."""
    obj = case[0]
    FLAGS, model, smc, obs, noise = _setup(*case, seed=5)
    _, ref0 = Hh.run_oracle(model, FLAGS, obj, obs, noise)
    teacher = {"idx_f": ref0["idx_f"]} if ref0["idx_f"] is not None else {}
    if obj in ("PSVO", "PSVOwR"):
        teacher["idx_b"] = ref0["idx_b"]
    if obj == "PSVOwR":
        teacher["idx_r"] = ref0["idx_r"]
    z_ref, P = _oracle_grads(model, FLAGS, obj, obs, noise, teacher)
    nz = Hh.noise_to_hip({**noise, **teacher}, "cuda")
    if "idx_f" in nz:
        nz.pop("u_f", None)
    if "sel_b" in nz:
        nz.pop("u_b", None)
    if "anc_r" in nz:
        nz.pop("u_r", None)
    def hip_pass():
        model.zero_grad()
        z, _ = smc.get_log_ZSMC(obs.float().cuda(), None, noise=nz)
        z.backward()
        torch.cuda.synchronize()
        return z
    z = hip_pass()
    assert abs(float(z.detach()) - float(z_ref)) <= 1e-4 * abs(float(z_ref))
    _check_grads(model, P, rerun=hip_pass)


def test_cross_lane_primitives(built_lib):
    """DPP / v_permlane*_swap helpers of csrc/common.h against their definitions"""
    import ctypes
    from psvo_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(0)
    x = torch.randn(64, generator=g)
    xin, out = x.cuda(), torch.empty(9 * 64, device="cuda")
    st = lib.psvo_selftest_lanes(ctypes.c_void_p(xin.data_ptr()), ctypes.c_void_p(out.data_ptr()),
                                 ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    _lib.check(st, "psvo_selftest_lanes")
    out = out.cpu().view(9, 64)
    lanes = torch.arange(64)
    for r, m in enumerate((1, 2, 4, 8, 16, 32)):
        assert torch.equal(out[r], x[lanes ^ m]), "xor_lane<%d>" % m
    assert torch.allclose(out[6], torch.cumsum(x.double(), 0).float(), atol=1e-5)
    assert torch.allclose(out[7], x.double().sum().float().expand(64), atol=1e-5)
    assert torch.equal(out[8], x.max().expand(64))


def test_cross_lane_primitives2(built_lib):
    """swap-add stages (v_permlane32_swap / v_permlane16_swap), the 16-lane row sum and the operand / accumulator layout of
    v_mfma_f32_16x16x4_f32 that csrc/bsim_bwd2_impl.h relies on, against their definitions"""
    import ctypes
    from psvo_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(3, 64, generator=g)
    xin, out = x.cuda().contiguous(), torch.empty(5 * 64, device="cuda")
    st = lib.psvo_selftest_lanes2(ctypes.c_void_p(xin.data_ptr()), ctypes.c_void_p(out.data_ptr()),
                                  ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    _lib.check(st, "psvo_selftest_lanes2")
    out = out.cpu().view(5, 64)
    lo, hi, ex = x[0], x[1], x[2]
    lanes = torch.arange(64)
    for row, mask in ((0, 32), (1, 16)):
        low_side = (lanes & mask) == 0
        want = torch.where(low_side, lo + lo[lanes ^ mask], hi + hi[lanes ^ mask])
        assert torch.equal(out[row], want), "swap_add%d" % mask
    assert torch.allclose(out[2], lo.view(4, 16).sum(1, keepdim=True).expand(4, 16).reshape(64), atol=1e-5)
    # D = A1 B + A2 B with A[i][k] in lane 16 k + i, B[k][j] in lane 16 k + j, D[i][j] in lane 16 (i // 4) + j, register i % 4
    A1, A2, Bm = lo.view(4, 16).t().double(), ex.view(4, 16).t().double(), hi.view(4, 16).double()
    D = (A1 + A2) @ Bm                                     # (16 i, 16 j)
    for reg in (0, 1):
        want = torch.stack([D[4 * (l // 16) + reg, l % 16] for l in range(64)]).float()
        assert torch.allclose(out[3 + reg], want, atol=1e-5), "mfma 16x16x4 register %d" % reg


BSIM_BWD_VARIANT_CASES = [c for c in CASES if c[0] == "PSVO"] + [("PSVO", 2, 9, 128, 16, 2, 1, 32, True, True),
                                                                ("PSVO", 1, 6, 200, 8, 3, 1, 32, True, True),
                                                                ("PSVO", 1, 5, 300, 32, 4, 2, 16, True, True)]


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("case", BSIM_BWD_VARIANT_CASES, ids=lambda c: "-".join(map(str, c)))
def test_bsim_backward_variants(built_lib, case, variant):
    """psvo_bsim_backward under every PSVO_TUNE_BSIM_BWD setting -- v1 (lane = (chain, half, m), per-j butterflies), v2
    ("j on lanes", per-j sums in registers + swap-add), v2 with the per-j sums on v_mfma_f32_16x16x4_f32 and v2 with the pair
    exponents on v_mfma_f32_16x16x4_f32 (Dx = 2) -- every gradient against the fp64 oracle's autograd (teacher-forced
    indices), same tolerance for all four"""
    from psvo_amd import _lib
    lib = _lib.load()
    obj = case[0]
    FLAGS, model, smc, obs, noise = _setup(*case, seed=17)
    _, ref0 = Hh.run_oracle(model, FLAGS, obj, obs, noise)
    teacher = {"idx_f": ref0["idx_f"], "idx_b": ref0["idx_b"]}
    z_ref, P = _oracle_grads(model, FLAGS, obj, obs, noise, teacher)
    nz = Hh.noise_to_hip({**noise, **teacher}, "cuda")
    nz.pop("u_f", None); nz.pop("u_b", None)
    assert lib.psvo_set_tuning(_lib.PSVO_TUNE_BSIM_BWD, variant) == 0
    try:
        model.zero_grad()
        z, _ = smc.get_log_ZSMC(obs.float().cuda(), None, noise=nz)
        z.backward()
        torch.cuda.synchronize()
    finally:
        lib.psvo_set_tuning(_lib.PSVO_TUNE_BSIM_BWD, -1)
    assert abs(float(z.detach()) - float(z_ref)) <= 1e-4 * abs(float(z_ref))
    _check_grads(model, P)


@pytest.mark.parametrize("variant", [1, 3, 4])
@pytest.mark.parametrize("sigma_f", [1.0, 0.3, 0.1])
def test_bsim_backward_small_transition_scale(built_lib, variant, sigma_f):
    """The exponent MFMA of variant 3 forms W' - |x' - F'|^2 as an expanded product (2 x'.F' - |x'|^2 - |F'|^2), which
    cancels where the differenced VALU form does not; the scaled coordinates grow as 1 / sigma_f.  Gradients at the
    reference's floor sigma_f = 1 and at 0.3 / 0.1 of it (f_sigma_min lowered) against the fp64 oracle; the VALU variant runs
    beside it as the control."""
    from psvo_amd import _lib
    import math
    lib = _lib.load()
    case = ("PSVO", 2, 8, 128, 16, 2, 1, 32, True, True)
    FLAGS, model, smc, obs, noise = _setup(*case, seed=19, q1_sigma_min=0.01, f_sigma_min=0.01)
    with torch.no_grad():        # transition scale (f == q1 under use_bootstrap): softplus(raw) = sigma_f
        model.q1_dist.sigma_con.fill_(math.log(math.expm1(sigma_f)))
    obs = O.fhn_synthetic(2, 8, seed=5)[1]
    _, ref0 = Hh.run_oracle(model, FLAGS, "PSVO", obs, noise)
    teacher = {"idx_f": ref0["idx_f"], "idx_b": ref0["idx_b"]}
    z_ref, P = _oracle_grads(model, FLAGS, "PSVO", obs, noise, teacher)
    assert abs(float(O.get_sigma(P["q1"])[0]) - sigma_f) < 1e-6
    nz = Hh.noise_to_hip({**noise, **teacher}, "cuda")
    nz.pop("u_f", None); nz.pop("u_b", None)
    assert lib.psvo_set_tuning(_lib.PSVO_TUNE_BSIM_BWD, variant) == 0
    try:
        model.zero_grad()
        z, _ = smc.get_log_ZSMC(obs.float().cuda(), None, noise=nz)
        z.backward()
        torch.cuda.synchronize()
    finally:
        lib.psvo_set_tuning(_lib.PSVO_TUNE_BSIM_BWD, -1)
    assert abs(float(z.detach()) - float(z_ref)) <= 1e-4 * abs(float(z_ref))
    _check_grads(model, P)


_variant_ids = lambda c: "-".join(map(str, c)) if isinstance(c, tuple) else ",".join("%s=%s" % kv for kv in c.items())


@pytest.mark.parametrize("case,extra", ENCODER_CASES + EMISSION_CASES + WIDTH_CASES, ids=_variant_ids)
def test_encoder_variants(built_lib, case, extra):
    """use_stack_rnn=False (two MultiRNNCells), BSim_use_single_RNN (forward cells only), poisson_emission
    (psvo_desc.emission = 1) and odd / mixed hidden widths: values, indices and every gradient against the oracle."""
    _variant_against_oracle(case, extra)


@pytest.mark.parametrize("case,extra", DEPTH_CASES, ids=_variant_ids)
def test_two_hidden_layers(built_lib, case, extra):
    """two hidden layers per particle MLP (psvo_desc.layers = 2; DEPTH_CASES): values, indices and every gradient
    against the oracle, in all five objectives."""
    _variant_against_oracle(case, extra)


# output_cov and diag_cov (src/runner_flag.py:67-70): every MLP carries the sigma_layer head and every scale is state-dependent
# (src/transformation/MLP.py:40-46,58-61, src/distribution/mvn.py:66-71) -- psvo_filter_forward_cov / _backward_cov.  All four
# bootstrap / 2q wirings, the three forward-filter objectives, both kernel sizes (<= 256 and 512 threads), padded widths and
# the Poisson emission (which drops MLP_g's head, src/distribution/poisson.py:33).
_COV = dict(output_cov=True, diag_cov=True)
COV_CASES = [
    (("AESMC", 2, 6, 16, 1, 2, 1, 32, True, True), _COV),
    (("AESMC", 2, 7, 100, 1, 3, 2, 32, False, True), _COV),
    (("AESMC", 2, 7, 130, 1, 2, 1, 16, True, False), _COV),
    (("AESMC", 2, 5, 300, 1, 4, 1, 64, False, False), _COV),
    (("AESMC", 2, 5, 512, 1, 4, 2, 32, False, True), _COV),
    (("IWAE", 2, 8, 96, 1, 3, 1, 16, False, True), _COV),
    (("IWAE", 1, 20, 4, 1, 2, 1, 32, True, True), _COV),
    (("SVO", 2, 8, 32, 1, 2, 1, 32, True, True), _COV),
    (("SVO", 2, 6, 16, 1, 3, 1, 32, False, False), _COV),
    (("SVO", 2, 6, 24, 1, 2, 2, 32, True, True), dict(_COV, q1_layers="24", g_layers="16", q0_layers="20", q2_layers="50")),
    (("AESMC", 2, 6, 16, 1, 2, 1, 32, True, True), dict(_COV, poisson_emission=True)),
    (("SVO", 2, 5, 40, 1, 3, 2, 16, False, True), dict(_COV, poisson_emission=True)),
    # the backward simulation on per-particle transition scales (psvo_bsim_forward_cov / _backward_cov)
    (("PSVO", 2, 6, 8, 4, 2, 1, 16, True, True), _COV),
    (("PSVO", 2, 7, 64, 16, 2, 1, 32, True, True), _COV),
    (("PSVO", 2, 6, 50, 8, 3, 1, 32, True, True), _COV),
    (("PSVO", 1, 5, 36, 32, 4, 2, 32, False, True), _COV),
    (("PSVO", 2, 5, 130, 16, 2, 1, 64, True, False), _COV),
    (("PSVO", 1, 5, 20, 4, 3, 1, 32, False, False), _COV),
    (("PSVO", 2, 6, 16, 8, 2, 1, 32, True, True), dict(_COV, poisson_emission=True)),
    (("PSVO", 2, 6, 16, 8, 2, 1, 32, False, True), dict(_COV, q1_layers="24", f_layers="40", g_layers="16", q0_layers="20",
                                                        q2_layers="50")),
    (("PSVO", 2, 6, 16, 8, 2, 1, 32, True, True), dict(_COV, BSim_use_single_RNN=True, y_smoother_Dhs="8,16")),
    # ... and with resampling across the chains (psvo_bsimwr_forward_cov / _backward_cov: one launch per time step)
    (("PSVOwR", 2, 6, 8, 4, 2, 1, 16, True, True), _COV),
    (("PSVOwR", 2, 7, 64, 16, 2, 1, 32, True, True), _COV),
    (("PSVOwR", 2, 6, 50, 8, 3, 1, 32, True, True), _COV),
    (("PSVOwR", 1, 5, 36, 32, 4, 2, 32, False, True), _COV),
    (("PSVOwR", 2, 5, 130, 16, 2, 1, 64, True, False), _COV),
    (("PSVOwR", 1, 5, 20, 4, 3, 1, 32, False, False), _COV),
    (("PSVOwR", 2, 5, 12, 4, 2, 1, 16, True, True), dict(_COV, poisson_emission=True)),
]


@pytest.mark.parametrize("case,extra", COV_CASES, ids=_variant_ids)
def test_state_dependent_scales(built_lib, case, extra):
    """output_cov and diag_cov: values, free-running indices and every gradient (both heads of every MLP, sigma_con, the
    hoisted networks and the encoder behind them) against the oracle."""
    _variant_against_oracle(case, extra)


@pytest.mark.parametrize("obj", ["PSVO", "PSVOwR", "SVO"])
def test_state_dependent_scales_long_sequence(built_lib, obj):
    """T = 120 under output_cov (teacher-forced indices): the reverse passes carry d mean / d scale through 120 scatter steps and
    accumulate 120 steps of atomics -- ELBO and every gradient against the fp64 oracle"""
    case = (obj, 2, 120, 24, 4, 2, 1, 32, True, True)
    FLAGS, model, smc, obs, noise = _setup(*case, seed=13, **_COV)
    _, ref0 = Hh.run_oracle(model, FLAGS, obj, obs, noise)
    teacher = {"idx_f": ref0["idx_f"]}
    if obj in ("PSVO", "PSVOwR"):
        teacher["idx_b"] = ref0["idx_b"]
    if obj == "PSVOwR":
        teacher["idx_r"] = ref0["idx_r"]
    z_ref, P = _oracle_grads(model, FLAGS, obj, obs, noise, teacher)
    nz = Hh.noise_to_hip({**noise, **teacher}, "cuda")
    for k in ("u_f", "u_b", "u_r"):
        nz.pop(k, None)

    def hip_pass():
        model.zero_grad()
        zz, _ = smc.get_log_ZSMC(obs.float().cuda(), None, noise=nz)
        zz.backward()
        torch.cuda.synchronize()
        return zz
    zz = hip_pass()
    assert abs(float(zz.detach()) - float(z_ref)) <= 1e-4 * abs(float(z_ref))
    _check_grads(model, P, rerun=hip_pass)


def test_state_dependent_scales_refusals(built_lib):
    """what is NOT built says so: two hidden layers with output_cov, the full-covariance form"""
    from psvo_amd.transformation.MLP import MLP_transformation
    with pytest.raises(NotImplementedError):
        MLP_transformation([8], 2, 2, output_cov=True, diag_cov=False)
    FLAGS, model, smc, obs, noise = _setup("AESMC", 2, 5, 8, 1, 2, 1, 32, True, True, **dict(_COV, q1_layers="32,32",
                                                                                            g_layers="32,32"))
    with pytest.raises(ValueError):
        smc.get_log_ZSMC(obs.float().cuda(), None, noise=Hh.noise_to_hip(noise, "cuda"))


def _variant_against_oracle(case, extra):
    obj = case[0]
    FLAGS, model, smc, obs, noise = _setup(*case, seed=7, **extra)
    z_free, ref0 = Hh.run_oracle(model, FLAGS, obj, obs, noise)
    with torch.no_grad():
        z, log = smc.get_log_ZSMC(obs.float().cuda(), None, noise=Hh.noise_to_hip(noise, "cuda"))
    assert abs(float(z) - float(z_free)) <= 1e-4 * abs(float(z_free))
    assert torch.allclose(log["Xs"].double().cpu(), ref0["Xs"], atol=2e-4, rtol=1e-5)
    teacher = {"idx_f": ref0["idx_f"]} if ref0["idx_f"] is not None else {}
    if obj in ("PSVO", "PSVOwR"):
        teacher["idx_b"] = ref0["idx_b"]
    if obj == "PSVOwR":
        teacher["idx_r"] = ref0["idx_r"]
    z_ref, P = _oracle_grads(model, FLAGS, obj, obs, noise, teacher)
    nz = Hh.noise_to_hip({**noise, **teacher}, "cuda")
    for k in ("u_f", "u_b", "u_r"):
        nz.pop(k, None)

    def hip_pass():
        model.zero_grad()
        zz, _ = smc.get_log_ZSMC(obs.float().cuda(), None, noise=nz)
        zz.backward()
        torch.cuda.synchronize()
        return zz
    zz = hip_pass()
    assert abs(float(zz.detach()) - float(z_ref)) <= 1e-4 * abs(float(z_ref))
    _check_grads(model, P, rerun=hip_pass)


@pytest.mark.parametrize("obj,T,N", [("PSVO", 2, 12), ("PSVO", 3, 130), ("PSVOwR", 2, 12), ("PSVOwR", 3, 33),
                                     ("AESMC", 1, 9), ("IWAE", 2, 7), ("SVO", 2, 128)])
def test_short_sequences(built_lib, obj, T, N):
    """T = 1 .. 3: the first / last steps of the persistent kernels are special-cased (t = 0 proposal and filter
    term, t = T-1 backward proposal, tile prefetch two steps ahead); values and gradients against the oracle."""
    case = (obj, 2, T, N, 8, 2, 1, 32, True, True)
    FLAGS, model, smc, obs, noise = _setup(*case, seed=11)
    z_free, ref0 = Hh.run_oracle(model, FLAGS, obj, obs, noise)
    with torch.no_grad():
        z, log = smc.get_log_ZSMC(obs.float().cuda(), None, noise=Hh.noise_to_hip(noise, "cuda"))
    assert abs(float(z) - float(z_free)) <= 1e-4 * abs(float(z_free))
    assert torch.allclose(log["Xs"].double().cpu(), ref0["Xs"], atol=2e-4, rtol=1e-5)
    teacher = {"idx_f": ref0["idx_f"]} if ref0["idx_f"] is not None else {}
    if obj in ("PSVO", "PSVOwR"):
        teacher["idx_b"] = ref0["idx_b"]
    if obj == "PSVOwR":
        teacher["idx_r"] = ref0["idx_r"]
    z_ref, P = _oracle_grads(model, FLAGS, obj, obs, noise, teacher)
    nz = Hh.noise_to_hip({**noise, **teacher}, "cuda")
    for k, v in (("u_f", "idx_f"), ("u_b", "sel_b"), ("u_r", "anc_r")):
        if v in nz:
            nz.pop(k, None)

    def hip_pass():
        model.zero_grad()
        zz, _ = smc.get_log_ZSMC(obs.float().cuda(), None, noise=nz)
        zz.backward()
        torch.cuda.synchronize()
        return zz
    zz = hip_pass()
    assert abs(float(zz.detach()) - float(z_ref)) <= 1e-4 * abs(float(z_ref))
    _check_grads(model, P, rerun=hip_pass)


def test_psvowr_step_replays_from_hipgraph(built_lib):
    """The PSVOwR kernels are cooperative launches (a cluster of workgroups per sequence exchanges data through HBM).
    Captured into a hipGraph and replayed, the local step must give the eagerly issued one's value bit for bit and its
    gradients to rounding (the PSVOwR reverse kernel's cross-chain scatter-add has a fixed order since round 2, but the
    filter's reverse pass still scatter-adds the resampling gather's gradient with LDS float atomics, whose order is not
    fixed; fixed injected noise), replay after replay, and no exchange poll may time out."""
    from psvo_amd.graph import GraphedStep
    from psvo_amd.optim import FlatParams
    FLAGS, model, smc, obs, noise = _setup("PSVOwR", 4, 12, 64, 8, 2, 1, 32, True, True, seed=3)
    nz = Hh.noise_to_hip(noise, "cuda")
    flat = FlatParams(model)
    obs_c = obs.float().cuda()

    def local():
        flat.zero_grad()
        z, _ = smc.get_log_ZSMC(obs_c, None, noise=nz)
        z.backward()
        return z.detach()
    z_e = local().clone()
    g_e = flat.grad.clone()
    torch.cuda.synchronize()
    smc.check_exchange()
    assert torch.isfinite(z_e) and float(g_e.abs().max()) > 0
    step = GraphedStep(local)
    for _ in range(3):
        flat.grad.fill_(float("nan"))          # the replay must rewrite every gradient itself
        z_g = step()
        torch.cuda.synchronize()
        assert torch.equal(z_g, z_e)
        assert torch.isfinite(flat.grad).all()
        assert (flat.grad - g_e).abs().max() <= 1e-5 * float(g_e.abs().max())
    smc.check_exchange()


def test_padded_width_step_with_flat_gradients_and_hipgraph(built_lib):
    """Mixed hidden widths (kernels run on zero-padded copies, gradients return through autograd into the flat buffer's
    views instead of being accumulated there by the kernels): eager and hipGraph-replayed local steps agree, and the
    gradient equals the one of the SAME network stored at the kernel width (explicitly padded parameters)."""
    from psvo_amd.graph import GraphedStep
    from psvo_amd.optim import FlatParams
    extra = dict(q1_layers="24", g_layers="16")
    FLAGS, model, smc, obs, noise = _setup("PSVO", 2, 8, 16, 8, 2, 1, 32, True, True, seed=4, **extra)
    assert smc._kernel_width() == (32, True, 1)
    nz = Hh.noise_to_hip(noise, "cuda")
    flat = FlatParams(model)
    obs_c = obs.float().cuda()

    def local():
        flat.zero_grad()
        z, _ = smc.get_log_ZSMC(obs_c, None, noise=nz)
        z.backward()
        return z.detach()
    z_e = local().clone()
    g_e = flat.grad.clone()
    torch.cuda.synchronize()
    assert torch.isfinite(z_e) and float(g_e.abs().max()) > 0
    step = GraphedStep(local)
    for _ in range(2):
        flat.grad.fill_(float("nan"))
        z_g = step()
        torch.cuda.synchronize()
        assert torch.equal(z_g, z_e)
        assert (flat.grad - g_e).abs().max() <= 1e-5 * float(g_e.abs().max())
    # the same network at the kernel width: q1 / q1_inv 24 -> 32 and g 16 -> 32 with zero hidden units
    F2, model2, smc2, _, _ = _setup("PSVO", 2, 8, 16, 8, 2, 1, 32, True, True, seed=4)
    assert smc2._kernel_width() == (32, False, 1)
    sd = model.state_dict()
    with torch.no_grad():
        for k, v in model2.state_dict().items():
            src = sd[k]
            if v.shape == src.shape:
                v.copy_(src)
            else:
                v.zero_()
                v[tuple(slice(0, n) for n in src.shape)].copy_(src)
    model2.zero_grad()
    z2, _ = smc2.get_log_ZSMC(obs_c, None, noise=nz)
    z2.backward()
    torch.cuda.synchronize()
    assert abs(float(z2) - float(z_e)) <= 1e-6 * abs(float(z_e))
    g1 = {k: p.grad for k, p in model.named_parameters()}
    for k, p in model2.named_parameters():
        a = g1[k]
        if a is None or p.grad is None:          # (parameters this objective does not use: None, or zeros in the flat buffer)
            assert (a is None or float(a.abs().max()) == 0.0) and (p.grad is None or float(p.grad.abs().max()) == 0.0), k
            continue
        b = p.grad[tuple(slice(0, n) for n in a.shape)]
        assert (a - b).abs().max() <= 1e-5 * max(float(b.abs().max()), 1e-6), k


def test_psvowr_many_sequences(built_lib):
    """More sequences than compute units: one workgroup per sequence (cluster size 1), launched as an ordinary kernel
    (nothing is exchanged between workgroups, so the grid need not be resident at once, which a cooperative launch would
    insist on).  ELBO, trajectories and cross-chain ancestors against the oracle; no exchange poll may time out."""
    B = 600
    FLAGS, model, smc, obs, noise = _setup("PSVOwR", B, 3, 24, 4, 2, 1, 16, True, True, seed=13)
    from psvo_amd import ops
    assert ops._lib.load().psvo_bsimwr_blocks(B, 24, 4) == 1
    with torch.no_grad():
        z, log = smc.get_log_ZSMC(obs.float().cuda(), None, noise=Hh.noise_to_hip(noise, "cuda"))
    torch.cuda.synchronize()
    smc.check_exchange()
    assert torch.isfinite(z) and torch.isfinite(log["Xs"]).all()
    z_ref, ref = Hh.run_oracle(model, FLAGS, "PSVOwR", obs, noise)
    assert abs(float(z) - float(z_ref)) <= 1e-4 * abs(float(z_ref))
    assert torch.allclose(log["Xs"].double().cpu(), ref["Xs"], atol=2e-4, rtol=1e-5)
    assert (log["bsim"]["anc"].permute(0, 2, 1).cpu().long() == ref["idx_r"]).all()
