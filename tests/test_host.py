"""CPU tests of the host-side mirror: flag registry, SSM inventory / sharing, encoder and k-step
prediction against the oracle, data loading / generation, R-square, and the refusal to run the
hot path without a GPU (no CPU fallback)."""
import math
import os
import pickle

import numpy as np
import pytest
import torch

from oracle import psvo_oracle as O
from psvo_amd import flags as F
from psvo_amd.model import SSM
from tests import helpers as Hh


def test_flag_defaults_match_reference_and_parser_forms():
    fl = F.Flags()
    assert (fl.Dx, fl.Dy, fl.n_particles, fl.batch_size, fl.lr, fl.epoch, fl.seed) == (2, 1, 16, 1, 3e-3, 200, 2)
    assert fl.PSVO and not (fl.SVO or fl.AESMC or fl.IWAE or fl.PSVOwR)
    assert fl.n_particles_for_BSim_proposal == 16 and fl.q1_layers == "32" and fl.use_bootstrap and fl.use_2_q
    assert abs(fl.lr_reduce_factor - 2 ** -0.5) < 1e-12 and abs(fl.min_lr - 3e-4) < 1e-12
    p = F.parse_flags(["--Dx=3", "--noPSVO", "--AESMC", "--lr", "0.01", "--use_2_q=false", "--q1_layers=64,64"])
    assert (p.Dx, p.PSVO, p.AESMC, p.lr, p.use_2_q, p.q1_layers) == (3, False, True, 0.01, False, "64,64")
    with pytest.raises(ValueError):
        F.parse_flags(["--no_such_flag=1"])
    with pytest.raises(ValueError):
        F.Flags(bogus=1)


def test_ssm_inventory_and_sharing():
    m = SSM(Hh.make_flags("PSVO"))
    assert m.f_dist is m.q1_dist and m.f_tran is m.q1_tran                  # use_bootstrap: f == q1
    assert m.g_dist.sigma_init == m.f_sigma_init                            # g_sigma_init quirk (model.py:37)
    assert m.BSim_q2_tran.Din == 64 and m.q2_tran.Din == 1 and m.q0_tran.Din == 1
    assert not hasattr(m, "X0_transformer_kernel")
    n_params = sum(p.numel() for p in m.parameters())
    assert n_params == 22426                                                # 8 MLPs + 4 LSTM cells
    m2 = SSM(Hh.make_flags("SVO", use_bootstrap=False))
    assert m2.f_dist is not m2.q1_dist and m2.q2_tran.Din == 64 and m2.q0_tran.Din == 2
    assert tuple(m2.X0_transformer_kernel.shape) == (128, 2)
    # per-particle MLPs: one hidden layer -> 4 tensors, two -> 6 (hidden_0, mu_layer, hidden_1); three have no kernel
    assert len(SSM(Hh.make_flags("AESMC", q1_layers="32,32")).q1_tran.hip_params()) == 6
    with pytest.raises(ValueError):
        SSM(Hh.make_flags("AESMC", q1_layers="32,32,32")).q1_tran.hip_params()


def test_kernel_hidden_width_selection():
    """per-particle MLPs run at the next instantiated kernel width on zero-padded hidden units; wider than 64 raises"""
    from psvo_amd.SMC.PSVO import PSVO
    from psvo_amd.SMC.AESMC import AESMC
    F = Hh.make_flags("PSVO", q1_layers="24", f_layers="40", g_layers="16", use_bootstrap=False)
    smc = PSVO(SSM(F), F)
    assert smc._kernel_width() == (64, True, 1)
    W1, b1, W2, b2 = smc._mlp_params(smc.model.g_tran)
    assert tuple(W1.shape) == (F.Dx, 64) and tuple(b1.shape) == (64,) and tuple(W2.shape) == (64, F.Dy)
    assert float(W1[:, 16:].abs().max()) == 0.0 and float(W2[16:].abs().max()) == 0.0
    assert torch.equal(W1[:, :16], smc.model.g_tran.kernels[0]) and torch.equal(W2[:16], smc.model.g_tran.mu_kernel)
    assert smc._gbuf(smc.model.g_tran) is None            # padded copies: gradients return through autograd
    (W1.sum() + W2.sum() + b1.sum()).backward()
    assert tuple(smc.model.g_tran.kernels[0].grad.shape) == (F.Dx, 16)
    F = Hh.make_flags("AESMC")                                            # defaults: every MLP 32 wide, no padding
    smc = AESMC(SSM(F), F)
    assert smc._kernel_width() == (32, False, 1)
    assert smc._mlp_params(smc.model.q1_tran)[0] is smc.model.q1_tran.kernels[0]
    F = Hh.make_flags("AESMC", g_layers="100")
    with pytest.raises(ValueError):
        AESMC(SSM(F), F)._kernel_width()


def test_two_hidden_layers_per_particle_mlp():
    """`*_layers="a,b"` (reference src/runner_flag.py:50-57, src/transformation/MLP.py:24-38): psvo_desc.layers = 2 at the next
    two-layer kernel width (32 / 64), both layers zero-padded; mixed depths have no kernel; the flat buffer lays such an MLP
    out as [W1 | b1 | Wh | bh | W2 | b2], the layout psvo_mlp2_wgrad writes"""
    from psvo_amd.SMC.PSVO import PSVO
    from psvo_amd.SMC.AESMC import AESMC
    from psvo_amd.optim import FlatParams
    F = Hh.make_flags("PSVO", q1_layers="24,40", g_layers="16,8")
    m = SSM(F)
    smc = PSVO(m, F)
    assert smc._kernel_width() == (64, True, 2)
    W1, b1, W2, b2, Wh, bh = smc._mlp_params(m.g_tran)
    assert [tuple(t.shape) for t in (W1, b1, W2, b2, Wh, bh)] == [(F.Dx, 64), (64,), (64, F.Dy), (F.Dy,), (64, 64), (64,)]
    assert torch.equal(Wh[:16, :8], m.g_tran.kernels[1]) and float(Wh[16:].abs().max()) == 0.0
    assert float(Wh[:, 8:].abs().max()) == 0.0 and float(W2[8:].abs().max()) == 0.0 and torch.equal(W2[:8], m.g_tran.mu_kernel)
    x = torch.randn(5, F.Dx)
    ref = m.g_tran.transform(x)[0]
    pad = torch.relu(torch.relu(x @ W1 + b1) @ Wh + bh) @ W2 + b2
    assert torch.allclose(ref, pad, atol=1e-6)
    first, extra = smc._mlp_args(smc._mlp_params(m.q1_tran), None, smc._mlp_params(m.g_tran))
    assert len(first) == 12 and first[4:8] == [None] * 4 and len(extra) == 6 and extra[2:4] == [None, None]
    F = Hh.make_flags("AESMC", q1_layers="16,16", g_layers="16,16")       # (two layers: no 16-wide kernels)
    smc = AESMC(SSM(F), F)
    smc.batch_size, smc.time = 2, 5
    assert smc._kernel_width() == (32, True, 2) and smc._make_desc(1, 32).layers == 2
    F = Hh.make_flags("AESMC", q1_layers="32,32")                         # g keeps one hidden layer
    with pytest.raises(ValueError):
        AESMC(SSM(F), F)._kernel_width()
    F = Hh.make_flags("AESMC", q1_layers="32,32", g_layers="32,32")
    m = SSM(F)
    fp = FlatParams(m)
    t = m.g_tran
    n = sum(q.numel() for q in (t.kernels[0], t.biases[0], t.kernels[1], t.biases[1], t.mu_kernel, t.mu_bias))
    assert t._flat_grad.numel() == n == F.Dx * 32 + 32 + 32 * 32 + 32 + 32 * F.Dy + F.Dy
    assert t._flat_grad.data_ptr() == t.kernels[0].grad.data_ptr()
    assert t.mu_bias.grad.data_ptr() == t._flat_grad[n - F.Dy:].data_ptr()
    assert fp.numel == sum(q.numel() for q in m.parameters())


def test_poisson_emission_mirror_matches_oracle_on_cpu():
    """FLAGS.poisson_emission: g_dist is the reference's tf_poisson (src/model.py:153-155), a unit-scale normal around
    softplus(MLP_g(x)) + 1e-6 (src/distribution/poisson.py:33-38), with no scale variable"""
    from psvo_amd.distribution.poisson import tf_poisson
    FLAGS = Hh.make_flags("AESMC", poisson_emission=True, Dx=3, Dy=2)
    torch.manual_seed(5)
    m = Hh.perturb_(SSM(FLAGS))
    assert isinstance(m.g_dist, tf_poisson) and not hasattr(m.g_dist, "sigma_con")
    assert not any("g_dist.sigma" in n for n, _ in m.named_parameters())
    P = m.export_reference_layout(torch.float64)
    assert "sigma_raw" not in P["g"]
    og = O.OracleSVO(P, Hh.oracle_flags(FLAGS, "AESMC"), smooth_obs=False).g
    x, y = torch.randn(7, 5, 3) * 3, torch.randn(7, 5, 2)
    assert torch.allclose(m.g_dist.mean(x).double(), og.mean(x.double()), atol=1e-6)
    assert float(m.g_dist.mean(x).min()) > 0
    assert torch.allclose(m.g_dist.log_prob(x, y).double(), og.log_prob(x.double(), y.double()), atol=1e-5)
    # closed form: -0.5 |y - lambda|^2 - Dy/2 log(2 pi)
    lam = og.mean(x.double())
    want = -0.5 * ((y.double() - lam) ** 2).sum(-1) - math.log(2 * math.pi)
    assert torch.allclose(og.log_prob(x.double(), y.double()), want, atol=1e-12)
    assert torch.equal(m.g_dist.get_sigma(), torch.ones(2))
    m2 = SSM(FLAGS).load_reference_layout(P)
    assert torch.equal(m2.g_tran.mu_kernel, m.g_tran.mu_kernel)


def test_export_import_roundtrip_and_sigma():
    torch.manual_seed(0)
    a = Hh.perturb_(SSM(Hh.make_flags("PSVO", use_bootstrap=False)))
    b = SSM(Hh.make_flags("PSVO", use_bootstrap=False)).load_reference_layout(a.export_reference_layout())
    for pa, pb in zip(a.parameters(), b.parameters()):
        assert torch.equal(pa, pb)
    P = a.export_reference_layout(torch.float64)
    assert torch.allclose(a.q1_dist.get_sigma().double(), O.get_sigma(P["q1"]), atol=1e-6)


def test_encoder_and_kstep_prediction_match_oracle_on_cpu():
    torch.manual_seed(1)
    FLAGS = Hh.make_flags("SVO", n_particles=6, y_smoother_Dhs="8,4", X0_smoother_Dhs="8")
    m = Hh.perturb_(SSM(FLAGS))
    P = m.export_reference_layout(torch.float64)
    x = torch.randn(3, 9, 1)
    ref = O.stack_bidirectional_rnn(x.double(), P["bRNN"]["y_smoother"])
    assert torch.allclose(m.y_smoother(x).double(), ref, atol=1e-5)
    from psvo_amd.SMC.SVO import SVO
    smc = SVO(m, FLAGS)
    o = O.OracleSVO(P, Hh.oracle_flags(FLAGS, "SVO"))
    Xs = torch.randn(3, 9, 6, 2)
    yh, y = smc.n_step_prediction(4, Xs, x)
    yh_ref, y_ref = o.n_step_prediction(4, Xs.double(), x.double())
    assert len(yh) == 5 and [tuple(v.shape) for v in yh] == [(3, 9 - k, 1) for k in range(5)]
    for a_, b_ in zip(yh, yh_ref):
        assert torch.allclose(a_.double(), b_, atol=1e-5)
    assert torch.allclose(smc.get_nextX(Xs[:, :, 0]).double(), o.get_nextX(Xs[:, :, 0].double()), atol=1e-5)


def test_encoder_wirings_match_oracle_on_cpu():
    """use_stack_rnn=False (MultiRNNCell per direction) and BSim_use_single_RNN (forward cells only): the host
    encoder (CPU path) against the oracle's restatement, incl. the X0 feature widths"""
    from psvo_amd.model import SSM
    from psvo_amd.SMC.PSVO import PSVO
    from psvo_amd.SMC.SVO import SVO
    x = torch.randn(3, 9, 1)
    # two MultiRNNCells
    FLAGS = Hh.make_flags("SVO", Dx=2, Dy=1, time=9, batch_size=3, n_particles=4, use_stack_rnn=False,
                          y_smoother_Dhs="6,4", X0_smoother_Dhs="5,3")
    torch.manual_seed(3)
    m = Hh.perturb_(SSM(FLAGS))
    P = m.export_reference_layout(torch.float64)
    fw, bw = O.bidirectional_rnn(x.double(), P["bRNN"]["y_smoother"])
    assert torch.allclose(m.y_smoother(x).double(), torch.cat([fw, bw], -1), atol=1e-5)
    X0, enc = SVO(m, FLAGS).preprocess_obs_w_bRNN(x)
    X0_ref, enc_ref = O.OracleSVO(P, Hh.oracle_flags(FLAGS, "SVO")).preprocess_obs_w_bRNN(x.double())
    assert X0.shape == (3, 2 * 3) and torch.allclose(X0.double(), X0_ref, atol=1e-5)
    assert torch.allclose(enc.double(), torch.stack(enc_ref, 1), atol=1e-5)
    assert m.q0_tran.kernels[0].shape[0] == 2 * 3          # q0 reads the (B, 2 Dh) feature directly (bootstrap and 2q)
    # forward cells only
    FLAGS = Hh.make_flags("PSVO", Dx=2, Dy=1, time=9, batch_size=3, n_particles=4, BSim_use_single_RNN=True,
                          y_smoother_Dhs="6,4")
    torch.manual_seed(4)
    m = Hh.perturb_(SSM(FLAGS))
    assert len(m.y_smoother.bw) == 0 and m.BSim_q2_tran.kernels[0].shape[0] == 4
    P = m.export_reference_layout(torch.float64)
    _, enc = PSVO(m, FLAGS).BS_preprocess_obs(x)
    _, enc_ref = O.OraclePSVO(P, Hh.oracle_flags(FLAGS, "PSVO")).BS_preprocess_obs(x.double())
    assert enc.shape == (3, 9, 4) and torch.allclose(enc.double(), torch.stack(enc_ref, 1), atol=1e-5)
    m2 = SSM(FLAGS).load_reference_layout(P)
    assert torch.equal(m2.y_smoother.fw[1].kernel, m.y_smoother.fw[1].kernel)


def test_r_square_matches_oracle_restatement():
    from psvo_amd.trainer import trainer
    g = np.random.RandomState(0)
    y = [g.randn(12, 10 - k, 1) for k in range(3)]
    yh = [v + 0.1 * g.randn(*v.shape) for v in y]
    r = trainer.evaluate_R_square(None, yh, y)
    ref = O.evaluate_R_square([torch.tensor(v) for v in yh], [torch.tensor(v) for v in y])
    assert np.allclose(r, ref.numpy(), atol=1e-12) and (r > 0.9).all()


def test_plateau_schedule_semantics():
    """adjust_lr (reference src/trainer.py:244-270): counters restart when the best held-out ELBO moves; the learning rate
    drops by lr_reduce_factor (floored at min_lr) every lr_reduce_patience evaluations x print_freq without improvement;
    StopTraining when early_stop_patience is hit exactly"""
    from psvo_amd.trainer import StopTraining, trainer
    FLAGS = Hh.make_flags("PSVO", lr=1e-2, lr_reduce_factor=0.5, lr_reduce_patience=4, early_stop_patience=10, min_lr=2e-3)
    tr = trainer(SSM(FLAGS), None, FLAGS)
    tr.save_model, tr.log_ZSMC_tests = False, []
    lrs, stopped = [], None
    for i, v in enumerate([-10.0, -9.0, -9.5, -9.4, -9.3, -9.2, -9.1, -9.05]):     # best at evaluation 1, then a plateau
        tr.log_ZSMC_tests.append(v)
        try:
            tr.adjust_lr(i, print_freq=2)
        except StopTraining:
            stopped = i
            break
        lrs.append(tr.lr)
    # evaluations 2, 3 without improvement: 2 * 2 == lr_reduce_patience -> halve at evaluation 3, again at 5;
    # 5 evaluations * 2 == early_stop_patience -> stop at evaluation 6
    assert lrs == [1e-2, 1e-2, 1e-2, 5e-3, 5e-3, 2.5e-3] and stopped == 6 and tr.bestCost == 1
    tr.lr, tr.early_stop_count = 2.5e-3, 0
    tr.lr_reduce_count = 1
    tr.log_ZSMC_tests = [-9.0, -9.5, -9.4]
    tr.bestCost = 0
    tr.adjust_lr(2, print_freq=2)
    assert tr.lr == 2e-3                                     # floored at min_lr
    tr.log_ZSMC_tests.append(-8.0)                           # improvement: counters restart
    tr.adjust_lr(3, print_freq=2)
    assert (tr.bestCost, tr.early_stop_count, tr.lr_reduce_count) == (3, 0, 0)


def test_epoch_data_dir_layout():
    from psvo_amd.trainer import trainer
    FLAGS = Hh.make_flags("PSVO")
    tr = trainer(SSM(FLAGS), None, FLAGS)
    tr.init_data_saving("/tmp/x/rslts/fhn/run7/")
    assert tr.epoch_data_DIR == "/tmp/x/rslts/epoch_data/fhn/run7/" and tr.save_res
    assert tr.log_ZSMC_trains == [] and tr.saving_num == FLAGS.saving_num


def test_data_loader_formats(tmp_path):
    from psvo_amd.utils.data_loader import load_data
    d = {"Ytrain": np.zeros((6, 5)), "Yvalid": np.ones((2, 5)), "Ytest": np.ones((3, 5)), "Xtrue": np.zeros((9, 5, 2))}
    p = tmp_path / "datadict"
    pickle.dump(d, open(p, "wb"))
    ht, hs, ot, os_ = load_data(str(p), 2, False, False)
    assert ot.shape == (6, 5, 1) and os_.shape == (3, 5, 1) and ht.shape == (6, 5, 2) and hs.shape == (3, 5, 2)
    pickle.dump({"Ytrain": np.zeros((4, 5, 1)), "Yvalid": np.zeros((2, 5, 1))}, open(p, "wb"))
    ht, hs, ot, os_ = load_data(str(p), 3, False, False)
    assert ht.shape == (4, 5, 3) and not ht.any()
    with pytest.raises(ValueError):
        load_data(str(p), 3, False, True)
    pickle.dump({"Ytrain": np.zeros((4, 5, 1))}, open(p, "wb"))
    with pytest.raises(ValueError):
        load_data(str(p), 3, False, False)


def test_data_generator_shapes_and_dynamics():
    from psvo_amd.utils.data_generator import generate_dataset
    np.random.seed(0)
    ht, hs, ot, os_ = generate_dataset(3, 2, 20, model="fhn", Dy=1)
    assert ht.shape == (3, 20, 2) and os_.shape == (2, 20, 1)
    assert np.abs(ot[:, :, 0] - ht[:, :, 0]).std() < 0.2 and np.abs(ht).max() < 5
    # one FHN step agrees with the oracle's RK4 restatement
    hid, _ = O.fhn_synthetic(4, 3, seed=2)
    from psvo_amd.utils.data_generator import _fhn_step
    nxt = np.stack([_fhn_step(h.numpy(), (1.0, 0.95, 0.05, 1.0, 0.15)) for h in hid[:, 0]])
    assert np.allclose(nxt, hid[:, 1].numpy(), atol=1e-5)
    with pytest.raises(ValueError):
        generate_dataset(1, 1, 5, model="nope")


def test_hot_path_refuses_to_run_without_gpu():
    """no CPU fallback: CPU tensors are rejected by the op wrappers"""
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from psvo_amd.SMC.AESMC import AESMC
    FLAGS = Hh.make_flags("AESMC", n_particles=8)
    smc = AESMC(SSM(FLAGS), FLAGS)
    with pytest.raises((ValueError, RuntimeError)):
        smc.get_log_ZSMC(torch.zeros(2, 5, 1), None)


def test_result_directory_and_param_files_follow_the_reference_naming(tmp_path, monkeypatch):
    """rslts/<name>/D<yymmdd>_<HHMMSS>_np_.._seed_../ with param.json (str values) -- what the reference's notebooks look for
    (src/rslts_saving/rslts_saving.py:14-47, notebook output 'RLT_DIR: .../rslts/notebook/D191011_212920_np_16_t_200_bs_1_...')"""
    import json
    import re
    from psvo_amd.rslts_saving.rslts_saving import NumpyEncoder, create_RLT_DIR, save_experiment_param
    monkeypatch.chdir(tmp_path)
    params = {"np": 16, "t": 200, "bs": 1, "lr": 0.003, "epoch": 400, "seed": 0, "rslt_dir_name": "notebook"}
    d = create_RLT_DIR(params)
    assert d.endswith("/") and os.path.isdir(d)
    rel = d[len(str(tmp_path).replace("\\", "/")):]
    assert re.fullmatch(r"/rslts/notebook/D\d{6}_\d{6}_np_16_t_200_bs_1_lr_0\.003_epoch_400_seed_0/", rel), rel
    FLAGS = Hh.make_flags("PSVO", n_particles=16)
    save_experiment_param(d, FLAGS)
    rec = json.load(open(d + "param.json"))
    assert rec["n_particles"] == "16" and rec["PSVO"] == "True" and list(rec) == sorted(rec)
    s = json.dumps({"a": np.float32(1.5), "b": np.arange(3), "c": [np.int64(2)]}, cls=NumpyEncoder)
    assert json.loads(s) == {"a": 1.5, "b": [0, 1, 2], "c": [2]}


def test_runner_objective_switch():
    """exactly one objective flag (src/runner.py:67-81); none raises ValueError, two trip the assert"""
    from psvo_amd import runner
    from psvo_amd.SMC.PSVOwR import PSVOwR
    FLAGS = Hh.make_flags("PSVOwR", n_particles=8)
    assert isinstance(runner._objective(SSM(FLAGS), FLAGS), PSVOwR)
    FLAGS = Hh.make_flags("AESMC", n_particles=8)
    FLAGS.AESMC = False
    with pytest.raises(ValueError):
        runner._objective(SSM(FLAGS), FLAGS)
    FLAGS.AESMC = FLAGS.IWAE = True
    with pytest.raises(AssertionError):
        runner._objective(SSM(FLAGS), FLAGS)


def test_quiver_lattice_dictionary():
    """lattice_val_<epoch>.p (trainer.py:337-361): keys and shapes the reference's notebook reads back; the lattice spans
    the particle-mean trajectories' bounding box widened by 5 % and nextX = f.mean on it (oracle's get_nextX)"""
    from psvo_amd.trainer import trainer
    from psvo_amd.SMC.SVO import SVO
    torch.manual_seed(5)
    FLAGS = Hh.make_flags("SVO", n_particles=4)
    m = Hh.perturb_(SSM(FLAGS))
    tr = trainer(m, SVO(m, FLAGS), FLAGS)
    tr.saving_num = 3
    Xs = np.random.RandomState(0).randn(5, 7, 4, 2)
    d = tr.quiver_lattice(Xs)
    assert set(d) == {"X_trajs", "X", "nextX"}
    assert d["X_trajs"].shape == (3, 7, 2) and d["X"].shape == (25, 25, 2) and d["nextX"].shape == (25, 25, 2)
    assert np.allclose(d["X_trajs"], Xs[:3].mean(axis=2))
    lo, hi = d["X_trajs"].reshape(-1, 2).min(0), d["X_trajs"].reshape(-1, 2).max(0)
    assert np.allclose(d["X"][0, 0], lo - 0.05 * (hi - lo)) and np.allclose(d["X"][-1, -1], hi + 0.05 * (hi - lo))
    assert np.allclose(d["X"][0, :, 1], d["X"][0, 0, 1]) and np.allclose(d["X"][:, 0, 0], d["X"][0, 0, 0])  # meshgrid "xy"
    o = O.OracleSVO(m.export_reference_layout(torch.float64), Hh.oracle_flags(FLAGS, "SVO"))
    ref = o.get_nextX(torch.as_tensor(d["X"].reshape(-1, 2))).reshape(25, 25, 2).numpy()
    assert np.allclose(d["nextX"], ref, atol=1e-5)


def test_covariance_head_mirror_matches_oracle_on_cpu():
    """output_cov and diag_cov (MLP.py:40-46,58-61, mvn.py:66-71): the sigma_layer head exists on every MLP (bias 1.0),
    scale = sigma_con + 0.1 (exp(head) + 1e-6), export / import carries it, and the full-covariance form is refused"""
    from psvo_amd.transformation.MLP import MLP_transformation
    torch.manual_seed(3)
    FLAGS = Hh.make_flags("SVO", n_particles=4, output_cov=True, diag_cov=True, use_bootstrap=False)
    m = SSM(FLAGS)
    for tr in (m.q0_tran, m.q1_tran, m.q2_tran, m.f_tran, m.g_tran):
        assert tr.sigma_kernel.shape == tr.mu_kernel.shape and torch.equal(tr.sigma_bias, torch.ones_like(tr.sigma_bias))
        assert set(tr.get_variables()) >= {"sigma_layer/weights", "sigma_layer/bias"}
    Hh.perturb_(m)
    P = m.export_reference_layout(torch.float64)
    x = torch.randn(5, 3, 2)
    for name, d in (("q1", m.q1_dist), ("f", m.f_dist), ("g", m.g_dist)):
        mu, sig = d.mean_and_sigma(x)
        mu_ref, sig_ref = O.OracleMVN(P[name]).mean_and_sigma(x.double())
        assert torch.allclose(mu.double(), mu_ref, atol=1e-5) and torch.allclose(sig.double(), sig_ref, atol=1e-5)
        assert float((sig - d.get_sigma()).min()) > 0.0          # the head adds to sigma_con
        lp = d.log_prob(x, x + 0.3)
        assert torch.allclose(lp.double(), O.OracleMVN(P[name]).log_prob(x.double(), x.double() + 0.3), atol=1e-4)
        assert torch.equal(d.mean(x), mu)
    b = SSM(FLAGS).load_reference_layout(P)
    for pa, pb in zip(m.parameters(), b.parameters()):
        assert torch.equal(pa, pb)
    with pytest.raises(NotImplementedError):
        MLP_transformation([8], 2, 2, output_cov=True, diag_cov=False)
    assert MLP_transformation([8], 2, 2, output_cov=False, diag_cov=True).transform(x)[1] is None    # MLP.py:40: no head
