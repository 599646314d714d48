"""GPU parity at the sizes the small suite never reaches: the N > 128 kernels (lane = particle, binary-search CDF, un-split
reverse filter) at N = 256 / 512, and T = 200 runs of the persistent kernels (double-buffer parities, the two-slot PSVOwR
ring, fp32 accumulation of the chain score over 200 steps).  Tolerances as in tests/test_gpu_parity.py.

Free-running draws are verified one by one instead of being compared with a second free run (after the first index that
differs the two runs are different particle systems): the oracle is re-run with the kernel's own indices teacher-forced, every
value must then agree, and every index must be the oracle's inverse-CDF draw on the oracle's logits or lie within
EDGE_TOL of the CDF edge that separates the two (fp32 weights and an fp32 prefix sum against the fp64 oracle)."""
import pytest
import torch

from oracle import psvo_oracle as O
from tests import helpers as Hh
from tests.test_gpu_parity import _setup, _oracle_grads, _check_grads

pytestmark = pytest.mark.gpu

# |u*total - edge| / total for an index that differs from the fp64 draw: fp32 log-weights of magnitude 10..100 carry an
# absolute error of ~1e-5, i.e. the weights a relative one of ~1e-5, the N-term fp32 prefix sum adds sqrt(N) ulp
EDGE_TOL = 5e-6

LARGE = [
    # objective, B, T, N, M, Dx, Dy, H, bootstrap, two_q      (BASELINE C4: N = 256, Dx = 2; C5: N = 512, Dx = 4)
    ("PSVO", 2, 12, 256, 16, 2, 1, 32, True, True),
    ("PSVO", 2, 12, 512, 16, 4, 1, 32, True, True),
    ("PSVO", 2, 12, 512, 16, 2, 1, 32, True, True),
    ("PSVO", 2, 12, 256, 16, 4, 1, 32, True, True),
    ("AESMC", 2, 12, 512, 1, 2, 1, 32, True, True),
    ("PSVOwR", 2, 12, 256, 16, 2, 1, 32, True, True),
]


def _teacher(ref, obj):
    t = {"idx_f": ref["idx_f"]} if ref["idx_f"] is not None else {}
    if obj in ("PSVO", "PSVOwR"):
        t["idx_b"] = ref["idx_b"]
    if obj == "PSVOwR":
        t["idx_r"] = ref["idx_r"]
    return t


def _hip_noise(noise, teacher):
    nz = Hh.noise_to_hip({**noise, **teacher}, "cuda")
    for k, v in (("u_f", "idx_f"), ("u_b", "sel_b"), ("u_r", "anc_r")):
        if v in nz:
            nz.pop(k, None)
    return nz


def _compare(obj, log, ref, z, z_ref):
    filt = log["filter"]
    assert torch.allclose(Hh.part_to_ref(filt["X"]), ref["X_prevs"], atol=2e-4, rtol=1e-5)
    assert torch.allclose(Hh.part_to_ref(filt["Xanc"]), ref["X_ancestors"], atol=2e-4, rtol=1e-5)
    assert torch.allclose(Hh.w_to_ref(filt["logW"]), ref["log_Ws"], atol=5e-4, rtol=1e-5)
    if obj == "PSVO":
        bs = log["bsim"]
        assert torch.allclose(Hh.part_to_ref(bs["bwX"]), ref["bw_Xs"], atol=2e-4, rtol=1e-5)
        assert torch.allclose(Hh.w_to_ref(bs["flp"]), ref["f_log_probs"], atol=5e-4, rtol=1e-5)
        assert torch.allclose(Hh.w_to_ref(bs["glp"]), ref["g_log_probs"], atol=5e-4, rtol=1e-5)
        assert torch.allclose(Hh.w_to_ref(bs["Omega"]), ref["bw_log_Omegas"], atol=5e-4, rtol=1e-5)
    if obj == "PSVOwR":
        bs = log["bsim"]
        assert torch.allclose(Hh.part_to_ref(bs["bwXanc"]), ref["bw_X_ancestors"], atol=2e-4, rtol=1e-5)
        assert torch.allclose(Hh.w_to_ref(bs["bwW"]), ref["bw_log_W"], atol=5e-4, rtol=1e-5)
    assert torch.allclose(log["Xs"].double().cpu(), ref["Xs"], atol=2e-4, rtol=1e-5)
    assert abs(float(z) - float(z_ref)) <= 1e-4 * abs(float(z_ref))


@pytest.mark.parametrize("case", LARGE, ids=lambda c: "-".join(map(str, c)))
def test_large_n_teacher_forced(built_lib, case):
    """values and every gradient at N = 256 / 512 with the oracle's indices injected"""
    obj = case[0]
    FLAGS, model, smc, obs, noise = _setup(*case, seed=21)
    z0, ref = Hh.run_oracle(model, FLAGS, obj, obs, noise)
    teacher = _teacher(ref, obj)
    nz = _hip_noise(noise, teacher)
    with torch.no_grad():
        z, log = smc.get_log_ZSMC(obs.float().cuda(), None, noise=nz)
    torch.cuda.synchronize()
    _compare(obj, log, ref, z, z0)
    z_ref, P = _oracle_grads(model, FLAGS, obj, obs, noise, teacher)

    def hip_pass():
        model.zero_grad()
        zz, _ = smc.get_log_ZSMC(obs.float().cuda(), None, noise=nz)
        zz.backward()
        torch.cuda.synchronize()
        return zz
    zz = hip_pass()
    assert abs(float(zz.detach()) - float(z_ref)) <= 1e-4 * abs(float(z_ref))
    _check_grads(model, P, rerun=hip_pass)
    if obj == "PSVOwR":
        smc.check_exchange()


@pytest.mark.parametrize("case", LARGE, ids=lambda c: "-".join(map(str, c)))
def test_large_n_free_running_draws(built_lib, case, record_property):
    """in-kernel multinomial draws at N = 256 / 512: every index is the oracle's draw or sits on a CDF edge; the count of
    edge flips is reported (and is what explains a free-running trajectory that leaves the oracle's)"""
    obj = case[0]
    FLAGS, model, smc, obs, noise = _setup(*case, seed=23)
    with torch.no_grad():
        z, log = smc.get_log_ZSMC(obs.float().cuda(), None, noise=Hh.noise_to_hip(noise, "cuda"))
    torch.cuda.synchronize()
    z_ref, ref, st = Hh.replay_with_hip_indices(model, FLAGS, obj, obs, noise, log)
    print("draws=%d flipped=%d worst edge distance=%.3e" % (st["draws"], st["off"], st["worst"]))
    record_property("flipped_draws", st["off"])
    _compare(obj, log, ref, z, z_ref)
    assert st["worst"] <= EDGE_TOL, st
    assert st["off"] <= max(3, int(20 * EDGE_TOL * st["draws"])), st     # a flip needs u within the tolerance of an edge


@pytest.mark.parametrize("obj", ["PSVO", "PSVOwR"])
def test_long_T_teacher_forced(built_lib, obj):
    """T = 200 at the headline N = 128, M = 16: values and every gradient against the fp64 oracle (teacher-forced), then the
    free-running draws of the same run verified one by one"""
    B = 2
    case = (obj, B, 200, 128, 16, 2, 1, 32, True, True)
    FLAGS, model, smc, obs, noise = _setup(*case, seed=31)
    _, obs = O.fhn_synthetic(B, 200, seed=3)                # an FHN trajectory, as in the headline workload
    z0, ref = Hh.run_oracle(model, FLAGS, obj, obs, noise)
    teacher = _teacher(ref, obj)
    nz = _hip_noise(noise, teacher)
    with torch.no_grad():
        z, log = smc.get_log_ZSMC(obs.float().cuda(), None, noise=nz)
    torch.cuda.synchronize()
    _compare(obj, log, ref, z, z0)
    z_ref, P = _oracle_grads(model, FLAGS, obj, obs, noise, teacher)

    def hip_pass():
        model.zero_grad()
        zz, _ = smc.get_log_ZSMC(obs.float().cuda(), None, noise=nz)
        zz.backward()
        torch.cuda.synchronize()
        return zz
    zz = hip_pass()
    assert abs(float(zz.detach()) - float(z_ref)) <= 1e-4 * abs(float(z_ref))
    _check_grads(model, P, rerun=hip_pass)
    if obj == "PSVOwR":
        smc.check_exchange()
    # free-running draws over the 200 steps
    with torch.no_grad():
        z, log = smc.get_log_ZSMC(obs.float().cuda(), None, noise=Hh.noise_to_hip(noise, "cuda"))
    torch.cuda.synchronize()
    z_ref, ref, st = Hh.replay_with_hip_indices(model, FLAGS, obj, obs, noise, log)
    print("draws=%d flipped=%d worst edge distance=%.3e" % (st["draws"], st["off"], st["worst"]))
    _compare(obj, log, ref, z, z_ref)
    assert st["worst"] <= EDGE_TOL, st


def test_psvowr_collapsed_ancestry_gradients(built_lib):
    """Weight degeneracy: every chain's cross-chain ancestor is teacher-forced into ONE workgroup's range of chains for 60
    steps (cluster of 8 workgroups per sequence).  The other seven workgroups then own no parent, poll nothing in the
    reverse pass and are free to run ahead of the one that does all the scatter-adds -- the case in which a two-slot
    exchange ring lets them overwrite words that have not been read yet.  Values, every gradient, and both kernels'
    time-out flags."""
    from psvo_amd import ops
    case = ("PSVOwR", 2, 60, 128, 16, 2, 1, 32, True, True)
    FLAGS, model, smc, obs, noise = _setup(*case, seed=41)
    assert ops._lib.load().psvo_bsimwr_blocks(2, 128, 16) == 8
    _, ref0 = Hh.run_oracle(model, FLAGS, "PSVOwR", obs, noise)
    T, N, B = ref0["idx_r"].shape
    g = torch.Generator().manual_seed(9)
    idx_r = torch.randint(16, 32, (T, N, B), generator=g)             # chains 16..31 = workgroup 1 of 8
    idx_r[::7] = 21                                                    # ... and whole steps on a single chain
    teacher = {"idx_f": ref0["idx_f"], "idx_b": ref0["idx_b"], "idx_r": idx_r}
    z_ref, P = _oracle_grads(model, FLAGS, "PSVOwR", obs, noise, teacher)
    nz = _hip_noise(noise, teacher)

    def hip_pass():
        model.zero_grad()
        zz, log = smc.get_log_ZSMC(obs.float().cuda(), None, noise=nz)
        zz.backward()
        torch.cuda.synchronize()
        return zz
    zz = hip_pass()
    smc.check_exchange()                                               # sticky flag: forward and reverse kernels
    assert abs(float(zz.detach()) - float(z_ref)) <= 1e-4 * abs(float(z_ref))
    _check_grads(model, P, rerun=hip_pass)
    for _ in range(3):                                                 # the hand-off must hold launch after launch
        hip_pass()
    smc.check_exchange()
    _check_grads(model, P)
