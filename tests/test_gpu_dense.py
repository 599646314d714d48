"""psvo_dense_forward / _backward (csrc/dense.hip: Dense layers on v_mfma_f32_16x16x4_f32) against the same layers in plain
PyTorch fp32, and the hoisted networks of a model with several hidden layers / wide encoders through them: values
atol 2e-5 (relative to the output scale), gradients rel 2e-4; then a PSVO evaluation with `q0_layers = q2_layers = "64,64"`
and 256-wide encoder features against the fp64 oracle (values, trajectories, every gradient)."""
import pytest
import torch

from oracle import psvo_oracle as O
from tests import helpers as Hh
from tests.test_gpu_parity import _check_grads, _oracle_grads, _setup

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("R,Din,Dout,relu", [(6400, 64, 32, True), (6400, 256, 64, True), (77, 1, 64, True),
                                             (1000, 64, 2, False), (130, 100, 100, True), (64, 16, 16, False),
                                             (5000, 3, 129, True)])
def test_dense_layer_matches_torch(built_lib, R, Din, Dout, relu):
    from psvo_amd.autograd import DenseFunction
    g = torch.Generator().manual_seed(R + Din + Dout)
    X = torch.randn(R, Din, generator=g).cuda().requires_grad_(True)
    W = (torch.randn(Din, Dout, generator=g) / Din ** 0.5).cuda().requires_grad_(True)
    b = (0.3 * torch.randn(Dout, generator=g)).cuda().requires_grad_(True)
    dY = torch.randn(R, Dout, generator=g).cuda()
    Y = DenseFunction.apply(X, W, b, relu)
    ref = X @ W + b
    if relu:
        ref = torch.relu(ref)
    assert torch.allclose(Y, ref, atol=2e-5 * max(1.0, float(ref.abs().max())), rtol=1e-5)
    grads = torch.autograd.grad(Y, [X, W, b], dY)
    grads_ref = torch.autograd.grad(ref, [X, W, b], dY)
    for a_, b_ in zip(grads, grads_ref):
        assert (a_ - b_).abs().max() <= 2e-4 * max(1.0, float(b_.abs().max()))
    # no input gradient requested
    Y2 = DenseFunction.apply(X.detach(), W, b, relu)
    gw, = torch.autograd.grad(Y2, [W], dY)
    assert (gw - grads_ref[1]).abs().max() <= 2e-4 * max(1.0, float(grads_ref[1].abs().max()))


def test_multi_hidden_layer_hoisted_networks_against_oracle(built_lib):
    """q0 / q2 / (BSim_q2, BSim_q_init share q2_layers / q0_layers) with two hidden layers, per-particle nets with one"""
    case = ("PSVO", 2, 7, 32, 8, 2, 1, 32, True, True)
    extra = dict(q0_layers="64,48", q2_layers="64,64", y_smoother_Dhs="16", X0_smoother_Dhs="16")
    FLAGS, model, smc, obs, noise = _setup(*case, seed=37, **extra)
    assert len(model.q2_tran.Dhs) == 2 and len(model.q0_tran.Dhs) == 2 and len(model.q1_tran.Dhs) == 1
    _, ref0 = Hh.run_oracle(model, FLAGS, "PSVO", obs, noise)
    with torch.no_grad():
        z, log = smc.get_log_ZSMC(obs.float().cuda(), None, noise=Hh.noise_to_hip(noise, "cuda"))
    assert torch.allclose(log["Xs"].double().cpu(), ref0["Xs"], atol=2e-4, rtol=1e-5)
    teacher = {"idx_f": ref0["idx_f"], "idx_b": ref0["idx_b"]}
    z_ref, P = _oracle_grads(model, FLAGS, "PSVO", obs, noise, teacher)
    nz = Hh.noise_to_hip({**noise, **teacher}, "cuda")
    nz.pop("u_f", None); nz.pop("u_b", None)
    model.zero_grad()
    zz, _ = smc.get_log_ZSMC(obs.float().cuda(), None, noise=nz)
    zz.backward()
    torch.cuda.synchronize()
    assert abs(float(zz.detach()) - float(z_ref)) <= 1e-4 * abs(float(z_ref))
    _check_grads(model, P)


def test_wide_encoder_features_go_through_dense(built_lib):
    """y_smoother_Dhs = 64 gives 128-wide features (fused kernel), 2 x 64 x 2 = 256 for the X0 feature of SVO: dense path"""
    case = ("SVO", 2, 6, 16, 4, 2, 1, 32, True, True)
    FLAGS, model, smc, obs, noise = _setup(*case, seed=39, y_smoother_Dhs="64", X0_smoother_Dhs="64")
    assert model.q0_tran.Din == 256
    z_ref, ref0 = Hh.run_oracle(model, FLAGS, "SVO", obs, noise)
    teacher = {"idx_f": ref0["idx_f"]}
    z_ref, P = _oracle_grads(model, FLAGS, "SVO", obs, noise, teacher)
    nz = Hh.noise_to_hip({**noise, **teacher}, "cuda")
    nz.pop("u_f", None)
    model.zero_grad()
    zz, _ = smc.get_log_ZSMC(obs.float().cuda(), None, noise=nz)
    zz.backward()
    torch.cuda.synchronize()
    assert abs(float(zz.detach()) - float(z_ref)) <= 1e-4 * abs(float(z_ref))
    _check_grads(model, P)
