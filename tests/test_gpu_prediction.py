"""Row f2 on the GPU: k-step prediction, R-square and get_nextX (reference src/SMC/SVO.py:371-409,
src/trainer.py:322-335) through the native row-MLP kernels, against the fp64 oracle on the same trajectories."""
import numpy as np
import pytest
import torch

from oracle import psvo_oracle as O
from tests import helpers as Hh
from tests.test_gpu_parity import _setup

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("obj,Dx,Dy,H", [("PSVO", 2, 1, 32), ("SVO", 3, 2, 16), ("AESMC", 4, 1, 64)])
def test_n_step_prediction_and_r_square_match_oracle(built_lib, obj, Dx, Dy, H):
    from psvo_amd.trainer import trainer
    B, T, N, K = 5, 24, 32, 6
    FLAGS, model, smc, obs, noise = _setup(obj, B, T, N, 8, Dx, Dy, H, True, True, seed=29, MSE_steps=K)
    P = model.export_reference_layout(torch.float64)
    o = O.OBJECTIVES[obj](P, Hh.oracle_flags(FLAGS, obj))
    g = torch.Generator().manual_seed(3)
    Xs = torch.randn(B, T, N, Dx, generator=g, dtype=torch.float64) * 2.0
    y_hat_ref, y_ref = o.n_step_prediction(K, Xs, obs)
    with torch.no_grad():
        y_hat, y = smc.n_step_prediction(K, Xs.float().cuda(), obs.float().cuda())
    assert len(y_hat) == K + 1 and all(v.is_cuda for v in y_hat)
    for k in range(K + 1):
        assert tuple(y_hat[k].shape) == (B, T - k, Dy)
        assert torch.allclose(y_hat[k].double().cpu(), y_hat_ref[k], atol=2e-4, rtol=1e-5), k
        assert torch.allclose(y[k].double().cpu(), y_ref[k], atol=1e-6)
    r2 = trainer.evaluate_R_square(None, [v.cpu().numpy() for v in y_hat], [v.cpu().numpy() for v in y])
    r2_ref = O.evaluate_R_square(y_hat_ref, y_ref).numpy()
    assert np.allclose(r2, r2_ref, atol=1e-4, rtol=1e-4)
    nxt = smc.get_nextX(Xs[:, :, 0].float().cuda())
    assert torch.allclose(nxt.double().cpu(), o.get_nextX(Xs[:, :, 0]), atol=2e-4, rtol=1e-5)


def test_trainer_evaluate_matches_oracle_prediction_chain(built_lib):
    """trainer.evaluate(["log_ZSMC", "y_hat", "y", "Xs"]) on the GPU: the k-step predictions it returns are the oracle's
    prediction chain applied to the trajectories it returns (the draws of an evaluation are the kernels' own)"""
    from psvo_amd.trainer import trainer
    B, T, N, K = 4, 20, 16, 5
    FLAGS, model, smc, obs, noise = _setup("PSVO", B, T, N, 4, 2, 1, 32, True, True, seed=31, MSE_steps=K)
    smc.generator = torch.Generator(device="cuda").manual_seed(7)
    tr = trainer(model, smc, FLAGS)
    hid = np.zeros((2 * B, T, 2))
    obs2 = torch.cat([obs, obs.flip(0)]).numpy()
    z, y_hat, y, Xs = tr.evaluate(["log_ZSMC", "y_hat", "y", "Xs"], {tr.obs: obs2, tr.hidden: hid})
    assert z.shape == (2,) and Xs.shape == (2 * B, T, N, 2) and len(y_hat) == K + 1
    P = model.export_reference_layout(torch.float64)
    o = O.OBJECTIVES["PSVO"](P, Hh.oracle_flags(FLAGS, "PSVO"))
    y_hat_ref, y_ref = o.n_step_prediction(K, torch.tensor(Xs).double(), torch.tensor(obs2).double())
    for k in range(K + 1):
        assert np.allclose(y_hat[k], y_hat_ref[k].numpy(), atol=3e-4, rtol=1e-5)
        assert np.allclose(y[k], y_ref[k].numpy(), atol=1e-6)
    r2 = tr.evaluate_R_square(y_hat, y)
    assert np.allclose(r2, O.evaluate_R_square(y_hat_ref, y_ref).numpy(), atol=1e-4, rtol=1e-3)
