"""GPU tests against the committed golden fixtures (tests/golden/*.npz) and size-independent
properties at BASELINE.json's full problem sizes, plus the optimizer / trainer plumbing."""
import math
import os
import sys

import numpy as np
import pytest
import torch

from tests import helpers as Hh

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import make_golden as MG  # noqa: E402

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _objective(obj):
    from psvo_amd.SMC.AESMC import AESMC
    from psvo_amd.SMC.IWAE import IWAE
    from psvo_amd.SMC.PSVO import PSVO
    from psvo_amd.SMC.PSVOwR import PSVOwR
    from psvo_amd.SMC.SVO import SVO
    return {"SVO": SVO, "PSVO": PSVO, "AESMC": AESMC, "IWAE": IWAE, "PSVOwR": PSVOwR}[obj]


@pytest.mark.parametrize("name", sorted(MG.CASES))
def test_golden_vectors(built_lib, name):
    """HIP path vs the committed vectors: free-running indices, ELBO, trajectories, one gradient"""
    from psvo_amd.model import SSM
    obj, B, T, N, M, Dx, Dy, H, Dh, boot, two_q = MG.CASES[name]
    ref = np.load(os.path.join(GOLD, name + ".npz"))
    fl, P0, _, noise0 = MG.build(name)
    P = Hh.fill_from_npz("params", P0, ref)
    noise = Hh.fill_from_npz("noise", noise0, ref)
    obs = torch.as_tensor(ref["obs"])
    hs = str(H)
    FLAGS = Hh.make_flags(obj, Dx=Dx, Dy=Dy, n_particles=N, n_particles_for_BSim_proposal=M, batch_size=B, time=T,
                          q0_layers=hs, q1_layers=hs, q2_layers=hs, f_layers=hs, g_layers=hs,
                          y_smoother_Dhs=str(Dh), X0_smoother_Dhs=str(Dh), use_bootstrap=boot, use_2_q=two_q,
                          **MG.EXTRA.get(name, {}))
    model = SSM(FLAGS).load_reference_layout(P).cuda()
    smc = _objective(obj)(model, FLAGS)
    z, log = smc.get_log_ZSMC(obs.float().cuda(), None, noise=Hh.noise_to_hip(noise, "cuda"))
    z.backward()
    torch.cuda.synchronize()
    zr = float(ref["log_ZSMC"])
    assert abs(float(z.detach()) - zr) <= 1e-4 * abs(zr)
    assert np.allclose(log["Xs"].detach().double().cpu().numpy(), ref["out.Xs"], atol=2e-4)
    if "out.idx_f" in ref.files:
        assert (log["filter"]["idx"].permute(0, 2, 1).cpu().numpy() == ref["out.idx_f"]).all()
    if "out.idx_b" in ref.files:
        assert (log["bsim"]["sel"].permute(0, 2, 1).cpu().numpy() == ref["out.idx_b"]).all()
    if "out.idx_r" in ref.files:
        assert (log["bsim"]["anc"].permute(0, 2, 1).cpu().numpy() == ref["out.idx_r"]).all()
        assert np.allclose(Hh.w_to_ref(log["bsim"]["bwW"].detach()).numpy(), ref["out.bw_log_W"], atol=5e-4)
    g = model.q1_tran.kernels[0].grad.double().cpu().numpy()
    gr = ref["grad.q1.layers.0.0"]
    assert np.abs(g - gr).max() <= 2e-3 * max(np.abs(gr).max(), 1e-6) + 1e-6


@pytest.mark.parametrize("wl", ["C2", "C*", "C3", "C4", "C5"])
def test_full_size_properties(built_lib, wl):
    """BASELINE.json sizes, where the oracle is too slow: identities that hold at any size"""
    sys.path.insert(0, os.path.dirname(GOLD + "/../.."))
    import bench
    obj, B, T, N, Dx, Dy, M, H, Dh = bench.WORKLOADS[wl]
    FLAGS, model, smc = bench.build_objective(bench.WORKLOADS[wl], "cuda")
    smc.generator = torch.Generator(device="cuda").manual_seed(0)
    g = torch.Generator().manual_seed(1)
    obs = torch.randn(B, T, Dy, generator=g).cuda()
    with torch.no_grad():
        z, log = smc.get_log_ZSMC(obs, None)
    f = log["filter"]
    assert math.isfinite(float(z))
    # (1) per-step logsumexp written by the kernel == logsumexp of the weights it wrote
    assert torch.allclose(f["lse"], torch.logsumexp(f["logW"], dim=2), atol=2e-4)
    # (2) resampled particles are exactly the gathered pre-resampling particles
    idx = f["idx"].long().unsqueeze(2).expand(-1, -1, Dx, -1)
    assert torch.equal(f["Xanc"], torch.gather(f["X"], 3, idx))
    assert int(f["idx"].min()) >= 0 and int(f["idx"].max()) < N
    # (3) ancestors follow the weights: sum_n onehot(idx) / N ~ softmax(logW) (multinomial, N*T*B draws)
    w = torch.softmax(f["logW"], dim=2)
    cnt = torch.zeros_like(w).scatter_add_(2, f["idx"].long(), torch.ones_like(w)) / N
    assert float((cnt - w).abs().mean()) < 2.5 * float((w * (1 - w) / N).clamp_min(0).sqrt().mean())
    if obj == "PSVO":
        b = log["bsim"]
        # (4) score == sum_t (f + g - Omega); ELBO == mean_b logsumexp_n score - log N
        sc = (b["flp"] + b["glp"] - b["Omega"]).sum(0)
        assert torch.allclose(b["score"], sc, atol=2e-2, rtol=1e-4)
        z2 = (torch.logsumexp(b["score"], 1) - math.log(N)).mean()
        assert abs(float(z) - float(z2)) < 1e-3 * abs(float(z2))
        assert int(b["sel"].min()) >= 0 and int(b["sel"].max()) < M
        assert log["Xs"].shape == (B, T, N, Dx)
    else:
        assert abs(float(z) - float(f["lse"].sum(0).mean())) < 1e-3 * abs(float(z))


def test_adam_matches_tf_formula(built_lib):
    from psvo_amd.optim import FlatParams, TFAdam
    torch.manual_seed(0)
    lin = torch.nn.Linear(7, 5).cuda()
    flat = FlatParams(lin)
    opt = TFAdam(flat)
    p = flat.flat.double().cpu().clone()
    m = torch.zeros_like(p); v = torch.zeros_like(p)
    for t in range(1, 6):
        g = torch.randn(flat.numel)
        flat.grad.copy_(g.cuda())
        opt.step(3e-3, world_size=2)
        ge = -g.double() / 2                       # maximise, summed over 2 ranks
        m = 0.9 * m + 0.1 * ge; v = 0.999 * v + 0.001 * ge * ge
        lr_t = 3e-3 * math.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
        p = p - lr_t * m / (v.sqrt() + 1e-8)
        assert torch.allclose(flat.flat.double().cpu(), p, atol=1e-6)
    assert lin.weight.data_ptr() == flat.flat.data_ptr()


@pytest.mark.parametrize("objective,hipgraph", [("PSVO", "1"), ("PSVO", "0"), ("PSVOwR", "1")])
def test_trainer_improves_elbo(built_lib, tmp_path, monkeypatch, objective, hipgraph):
    """a few epochs of the mirrored trainer on a small FHN set: ELBO goes up, artefacts are written
    (local step replayed from a hipGraph -- PSVOwR's cooperative kernels included -- and issued eagerly)"""
    monkeypatch.setenv("PSVO_HIPGRAPH", hipgraph)
    from oracle import psvo_oracle as O
    from psvo_amd.model import SSM
    from psvo_amd.trainer import trainer
    PSVO = _objective(objective)
    monkeypatch.chdir(tmp_path)
    hid, obs = O.fhn_synthetic(12, 30, seed=0)
    FLAGS = Hh.make_flags(objective, n_particles=16, n_particles_for_BSim_proposal=4, batch_size=4, time=30, epoch=4,
                          lr=1e-2, MSE_steps=5, saving_num=4, rslt_dir_name="t")
    torch.manual_seed(0); np.random.seed(0)
    model = SSM(FLAGS).cuda()
    smc = PSVO(model, FLAGS)
    smc.generator = torch.Generator(device="cuda").manual_seed(1)
    tr = trainer(model, smc, FLAGS)
    rlt = str(tmp_path) + "/rslts/t/run/"
    os.makedirs(rlt)
    tr.init_data_saving(rlt)
    hist, log = tr.train(obs[:8].numpy(), obs[8:].numpy(), hid[:8].numpy(), hid[8:].numpy(), print_freq=1)
    assert len(hist["log_ZSMC_trains"]) == 5 and hist["log_ZSMC_trains"][-1] > hist["log_ZSMC_trains"][0] + 1.0
    assert hist["R_square_trains"][0].shape == (6,)
    assert os.path.exists(tr.epoch_data_DIR + "metric_4.p") and os.path.exists(tr.epoch_data_DIR + "trajectory_4.p")
    Xs = tr.evaluate(log["Xs"], tr.saving_feed_dict)
    assert Xs.shape == (4, 30, 16, 2)
    graphs = tr.__dict__.get("_graphs", {})
    assert (len(graphs) == 1 and all(g is not False for g in graphs.values())) if hipgraph == "1" else not graphs


@pytest.mark.parametrize("objective,hipgraph", [("PSVO", "1"), ("PSVO", "0"), ("AESMC", "1"), ("PSVOwR", "1")])
def test_trainer_with_state_dependent_scales(built_lib, tmp_path, monkeypatch, objective, hipgraph):
    """output_cov and diag_cov through the mirrored trainer (flat parameter buffer incl. the sigma_layer heads, Adam, the
    local step replayed from a hipGraph or issued eagerly): the ELBO goes up and the evaluation chain runs"""
    monkeypatch.setenv("PSVO_HIPGRAPH", hipgraph)
    from oracle import psvo_oracle as O
    from psvo_amd.model import SSM
    from psvo_amd.trainer import trainer
    cls = _objective(objective)
    monkeypatch.chdir(tmp_path)
    hid, obs = O.fhn_synthetic(12, 30, seed=0)
    FLAGS = Hh.make_flags(objective, n_particles=16, n_particles_for_BSim_proposal=4, batch_size=4, time=30, epoch=4,
                          lr=1e-2, MSE_steps=5, saving_num=4, rslt_dir_name="t", output_cov=True, diag_cov=True)
    torch.manual_seed(0); np.random.seed(0)
    model = SSM(FLAGS).cuda()
    smc = cls(model, FLAGS)
    smc.generator = torch.Generator(device="cuda").manual_seed(1)
    tr = trainer(model, smc, FLAGS)
    rlt = str(tmp_path) + "/rslts/t/run/"
    os.makedirs(rlt)
    tr.init_data_saving(rlt)
    heads0 = [tr_.sigma_kernel.detach().clone() for tr_ in (model.q1_tran, model.g_tran)]
    hist, log = tr.train(obs[:8].numpy(), obs[8:].numpy(), hid[:8].numpy(), hid[8:].numpy(), print_freq=1)
    assert len(hist["log_ZSMC_trains"]) == 5 and hist["log_ZSMC_trains"][-1] > hist["log_ZSMC_trains"][0] + 1.0
    assert all(np.isfinite(hist["log_ZSMC_tests"]))
    for w0, tr_ in zip(heads0, (model.q1_tran, model.g_tran)):       # the heads are trained
        assert float((tr_.sigma_kernel.detach() - w0).abs().max()) > 1e-4
    graphs = tr.__dict__.get("_graphs", {})
    assert (len(graphs) == 1 and all(g is not False for g in graphs.values())) if hipgraph == "1" else not graphs


@pytest.mark.parametrize("obj,layers", [("PSVO", "32"), ("SVO", "32"), ("PSVO", "32,32"), ("SVO", "64,64")])
def test_flat_buffer_gradients_match_autograd_path(built_lib, obj, layers):
    """with optim.FlatParams the native backward passes accumulate straight into the flat gradient buffer
    (and all sigmas come from one fused launch): same gradients as the plain autograd path -- one hidden layer per
    particle MLP ([W1|b1|W2|b2] slices) and two ([W1|b1|Wh|bh|W2|b2], psvo_mlp2_wgrad)"""
    from psvo_amd.model import SSM
    from psvo_amd.optim import FlatParams
    FLAGS = Hh.make_flags(obj, n_particles=32, n_particles_for_BSim_proposal=8, batch_size=3, time=9,
                          y_smoother_Dhs="8", X0_smoother_Dhs="8", use_bootstrap=(obj == "PSVO"),
                          q1_layers=layers, f_layers=layers, g_layers=layers)
    torch.manual_seed(0)
    m1 = Hh.perturb_(SSM(FLAGS)).cuda()
    m2 = SSM(FLAGS).cuda()
    m2.load_state_dict(m1.state_dict())
    obs = torch.randn(3, 9, 1, generator=torch.Generator().manual_seed(2)).cuda()
    outs = []
    for mdl, flat in ((m1, False), (m2, True)):
        smc = _objective(obj)(mdl, FLAGS)
        smc.generator = torch.Generator(device="cuda").manual_seed(5)
        fp = FlatParams(mdl) if flat else None
        for _ in range(2):                                   # twice: accumulation into a zeroed buffer each time
            if fp is not None:
                fp.zero_grad()
            else:
                mdl.zero_grad()
            smc.generator.manual_seed(5)
            z, _ = smc.get_log_ZSMC(obs, None)
            z.backward()
        torch.cuda.synchronize()
        outs.append((float(z.detach()), {n: (torch.zeros_like(p) if p.grad is None else p.grad.detach().clone())
                                         for n, p in mdl.named_parameters()}))
    assert abs(outs[0][0] - outs[1][0]) <= 1e-5 * abs(outs[0][0])
    for n, g in outs[0][1].items():
        g2 = outs[1][1][n]
        assert torch.allclose(g, g2, atol=2e-5 + 1e-4 * float(g.abs().max()), rtol=1e-3), n


@pytest.fixture
def wgrad2_variant(request):
    from psvo_amd import _lib
    lib = _lib.load()
    assert lib.psvo_set_tuning(_lib.PSVO_TUNE_WGRAD2, request.param) == 0
    yield request.param
    lib.psvo_set_tuning(_lib.PSVO_TUNE_WGRAD2, 0)


@pytest.mark.parametrize("wgrad2_variant", [0, 2, 3], indirect=True)
@pytest.mark.parametrize("shape,Din,H,Dout", [((3, 2, 2, 50, 4), 2, 32, 2), ((7, 3, 3, 130), 3, 64, 1), ((2, 5, 4, 9, 8), 4, 64, 4),
                                             ((40, 2, 2, 64, 16), 2, 64, 2), ((1, 1, 2, 5), 2, 32, 3)])
def test_mlp2_wgrad_matches_torch(built_lib, shape, Din, H, Dout, wgrad2_variant):
    """psvo_mlp2_wgrad (two hidden layers; the H x H products on v_mfma_f32_16x16x4_f32, or -- PSVO_TUNE_WGRAD2 = 2 / 3 -- on
    the bf16 matrix instructions with every operand split into two / three bf16 pieces) against torch autograd of the same
    MLP in fp64 over the same rows: every parameter gradient rel 1e-4 of its largest entry; ragged row counts (not a
    multiple of the 64-row tile), rows straddling segments, and accumulation into an existing buffer.
    The split variants are A/B knobs, not the default: with operands carried to ~2^-17 (two pieces) the second layer's
    pre-activations are off by ~1e-5, which flips the relu mask of ~1e-5 of the (row, unit) pairs -- on these random rows,
    whose gradient sums cancel to ~sqrt(rows), that is ~3e-3 of a tensor's largest entry; three pieces (~2^-24) flip a handful
    more masks than the f32 instruction does.  Their bars are set accordingly."""
    from psvo_amd import ops
    bar = {0: 1e-4, 3: 1e-3, 2: 2e-2}[wgrad2_variant]
    g = torch.Generator().manual_seed(sum(shape) + H)
    dshape = shape[:2] + (Dout,) + shape[3:]
    X = torch.randn(*shape, generator=g)
    dOut = torch.randn(*dshape, generator=g)
    W1 = torch.randn(Din, H, generator=g) / Din ** 0.5
    b1 = 0.3 * torch.randn(H, generator=g)
    Wh = torch.randn(H, H, generator=g) / H ** 0.5
    bh = 0.3 * torch.randn(H, generator=g)
    W2 = torch.randn(H, Dout, generator=g) / H ** 0.5
    b2 = 0.3 * torch.randn(Dout, generator=g)
    ps = [t.double().requires_grad_(True) for t in (W1, b1, Wh, bh, W2, b2)]
    rows = X.double().movedim(2, -1).reshape(-1, Din)
    out = torch.relu(torch.relu(rows @ ps[0] + ps[1]) @ ps[2] + ps[3]) @ ps[4] + ps[5]
    out.backward(dOut.double().movedim(2, -1).reshape(-1, Dout))
    ref = torch.cat([t.grad.reshape(-1) for t in ps])
    w = tuple(t.cuda() for t in (W1, b1, W2, b2, Wh, bh))
    got = ops.mlp_wgrad(X.cuda(), dOut.cuda(), w, Din, H, Dout)
    torch.cuda.synchronize()
    assert got.numel() == ref.numel() == ops.mlp_grad_size(Din, H, Dout, 2)
    off = 0
    for name, t in zip(("dW1", "db1", "dWh", "dbh", "dW2", "db2"), ps):
        a, b = got[off:off + t.numel()].double().cpu(), t.grad.reshape(-1)
        assert torch.allclose(a, b, atol=bar * float(b.abs().max()) + 1e-6, rtol=bar), name
        off += t.numel()
    acc = torch.ones_like(got)
    ops.mlp_wgrad(X.cuda(), dOut.cuda(), w, Din, H, Dout, grad=acc)
    torch.cuda.synchronize()
    assert torch.allclose(acc, got + 1.0, atol=1e-5 * float(got.abs().max()) + 1e-6, rtol=1e-5)
    g4 = ops.split_mlp_grad(got, Din, H, Dout, layers=2)
    assert [tuple(v.shape) for v in g4] == [(Din, H), (H,), (H, Dout), (Dout,), (H, H), (H,)]


@pytest.mark.parametrize("shape", [(3, 2, 0, 50, 4), (1, 1, 0, 5), (2, 1, 0, 5000), (40, 8, 0, 100, 16), (200, 8, 0, 128, 16), (7, 3, 0, 130)])
@pytest.mark.parametrize("H", [16, 32, 64])
def test_mlp_wgrad_kernels(built_lib, monkeypatch, shape, H):
    """psvo_mlp_wgrad (one hidden layer), every Din x Dout instantiation of the default kernel (the waves of a workgroup
    are row groups x block columns over shared rows, 16 or 8 hidden units per column) against (a) torch autograd of the same
    MLP in fp64 over the same rows, rel 1e-4 of a tensor's largest entry, and (b) the round-1 kernel (PSVO_WGRAD_OLD=1), which
    sums the same products in another order: rel 2e-5.
    Shapes: fewer rows than one workgroup, a segment longer than the grid stride, strides that are not a multiple of the
    segment length (the incremental (segment, row) advance), and 3.3e6 rows on the full 1024-workgroup grid; H = 16 / 32 /
    64 give 1 -- 8 block columns (1, 2 or 4 per workgroup)."""
    from psvo_amd import ops
    big = shape[0] * shape[1] * (shape[3] * (shape[4] if len(shape) > 4 else 1)) > 1_000_000
    for Din in (1, 2, 3, 4):
        for Dout in (1, 2, 3, 4):
            if big and (Din, Dout) not in ((2, 2), (2, 1), (4, 4), (3, 1)):
                continue
            g = torch.Generator().manual_seed(sum(shape) + H + 10 * Din + Dout)
            X = torch.randn(*(shape[:2] + (Din,) + shape[3:]), generator=g)
            dOut = torch.randn(*(shape[:2] + (Dout,) + shape[3:]), generator=g)
            W1 = torch.randn(Din, H, generator=g) / Din ** 0.5
            b1 = 0.3 * torch.randn(H, generator=g)
            W2 = torch.randn(H, Dout, generator=g) / H ** 0.5
            b2 = 0.3 * torch.randn(Dout, generator=g)
            w = tuple(t.cuda() for t in (W1, b1, W2, b2))
            Xc, dc = X.cuda(), dOut.cuda()
            monkeypatch.setenv("PSVO_WGRAD_OLD", "1")
            old = ops.mlp_wgrad(Xc, dc, w, Din, H, Dout).clone()
            monkeypatch.delenv("PSVO_WGRAD_OLD")
            got = ops.mlp_wgrad(Xc, dc, w, Din, H, Dout)
            torch.cuda.synchronize()
            assert torch.allclose(got, old, atol=2e-5 * float(old.abs().max()) + 1e-6, rtol=2e-5), (Din, Dout)
            ps = [t.double().cuda().requires_grad_(True) for t in (W1, b1, W2, b2)]
            rows = Xc.double().movedim(2, -1).reshape(-1, Din)
            # the relu mask as the kernel's f32 fma chain sees it (the exact product and sum in fp64, rounded once to f32):
            # among 5e7 (row, unit) pairs a handful have a pre-activation whose sign differs between f32 and fp64, and one
            # such row moves a sum by more than the bar
            pre = b1.cuda()[None, :].expand(rows.shape[0], H)
            for i in range(Din):
                pre = (rows[:, i:i + 1] * ps[0][i].detach()[None, :] + pre.double()).float()
            out = ((rows @ ps[0] + ps[1]) * (pre > 0)) @ ps[2] + ps[3]
            out.backward(dc.double().movedim(2, -1).reshape(-1, Dout))
            off = 0
            for name, t in zip(("dW1", "db1", "dW2", "db2"), ps):
                a, b = got[off:off + t.numel()].double(), t.grad.reshape(-1)
                assert torch.allclose(a, b, atol=1e-4 * float(b.abs().max()) + 1e-6, rtol=1e-4), (name, Din, Dout)
                off += t.numel()
            acc = torch.ones_like(got)
            ops.mlp_wgrad(Xc, dc, w, Din, H, Dout, grad=acc)
            torch.cuda.synchronize()
            assert torch.allclose(acc, got + 1.0, atol=1e-5 * float(got.abs().max()) + 1e-6, rtol=1e-5)


@pytest.mark.parametrize("R,Din,H,Dout", [(6400, 64, 32, 2), (32, 1, 32, 2), (77, 128, 64, 4), (1000, 3, 16, 1)])
def test_rows_mlp_matches_torch(built_lib, R, Din, H, Dout):
    """psvo_rows_mlp_forward / _backward (the hoisted q0 / q2 / BSim_q2 / BSim_q_init means) against the same MLP
    in plain PyTorch fp32: values atol 1e-5 (relative to the output scale), all gradients rel 1e-4."""
    from psvo_amd.autograd import RowsMLPFunction
    g = torch.Generator().manual_seed(R + Din)
    X = torch.randn(R, Din, generator=g).cuda().requires_grad_(True)
    W1 = (torch.randn(Din, H, generator=g) / Din ** 0.5).cuda().requires_grad_(True)
    b1 = (0.3 * torch.randn(H, generator=g)).cuda().requires_grad_(True)
    W2 = (torch.randn(H, Dout, generator=g) / H ** 0.5).cuda().requires_grad_(True)
    b2 = (0.3 * torch.randn(Dout, generator=g)).cuda().requires_grad_(True)
    dOut = torch.randn(R, Dout, generator=g).cuda()
    out = RowsMLPFunction.apply(None, X, W1, b1, W2, b2)
    ref = torch.relu(X @ W1 + b1) @ W2 + b2
    assert torch.allclose(out, ref, atol=1e-5 * max(1.0, float(ref.abs().max())), rtol=1e-5)
    grads = torch.autograd.grad(out, [X, W1, b1, W2, b2], dOut)
    grads_ref = torch.autograd.grad(ref, [X, W1, b1, W2, b2], dOut)
    for a_, b_ in zip(grads, grads_ref):
        assert (a_ - b_).abs().max() <= 1e-4 * max(1.0, float(b_.abs().max()))
    # accumulation into a given flat gradient slice, no dX requested
    NP = Din * H + H + H * Dout + Dout
    flat = torch.ones(NP, device="cuda")
    out2 = RowsMLPFunction.apply(flat, X.detach(), W1, b1, W2, b2)
    out2.backward(dOut)
    want = torch.cat([t.reshape(-1) for t in grads_ref[1:]]) + 1.0
    assert (flat - want).abs().max() <= 1e-4 * max(1.0, float(want.abs().max()))


@pytest.mark.parametrize("B,N", [(32, 128), (1, 8), (5, 100), (7, 1024)])
def test_elbo_bsim_mean_matches_torch(built_lib, B, N):
    """psvo_elbo_bsim_mean / _backward (reference PSVO.py:52-67 with its reduce_mean) against
    torch.logsumexp(...).mean() in fp32: value rel 1e-6, d/dscore within 2e-5 of its largest entry."""
    import math
    from psvo_amd import ops
    from psvo_amd.autograd import ElboBsimFunction
    g = torch.Generator().manual_seed(B * 1000 + N)
    score = (50.0 * torch.randn(B, N, generator=g) - 300.0).cuda().requires_grad_(True)
    desc = ops.make_desc(B, 3, N, 4, 2, 1, 16)
    z = ElboBsimFunction.apply(desc, score)
    ref = (torch.logsumexp(score, dim=1) - math.log(float(N))).mean()
    assert z.shape == ref.shape
    assert abs(float(z) - float(ref)) <= 1e-6 * abs(float(ref))
    (d,) = torch.autograd.grad(z, score, torch.tensor(0.7, device="cuda"))
    (dref,) = torch.autograd.grad(ref, score, torch.tensor(0.7, device="cuda"))
    assert (d - dref).abs().max() <= 2e-5 * float(dref.abs().max())
    # per-sequence entry point agrees with the fused mean
    per_seq = ops.elbo_bsim(desc, score.detach())
    assert abs(float(per_seq.mean()) - float(z)) <= 1e-6 * abs(float(ref))
