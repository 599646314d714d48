"""world_size-2 data-parallel path on CPU (gloo): sharding, the single flat gradient all-reduce,
replica consistency.  The per-rank compute is the CPU oracle here (checker only -- the HIP path
needs a GPU); what is under test is psvo_amd.dp / psvo_amd.optim.FlatParams."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import psvo_oracle as O


def _worker(rank, world, port, q):
    try:
        _worker_body(rank, world, port, q)
    except Exception as e:  # report instead of letting the parent wait for its timeout
        import traceback
        traceback.print_exc()
        q.put((rank, "error: %r" % (e,)))


def _sharded_evaluate_ok(rank, world):
    """trainer.evaluate deals the evaluation batches round-robin over the ranks and gathers: every rank must return what a
    single process returns (a stub objective stands in for the HIP path, which needs a GPU)"""
    import numpy as np
    from psvo_amd.flags import Flags
    from psvo_amd.trainer import trainer

    class StubModel(torch.nn.Linear):
        obs, hidden = "obs", "hidden"

    class StubSMC:
        calls = 0

        def get_log_ZSMC(self, obs, hidden):
            StubSMC.calls += 1
            Xs = (obs[:, :, None, :1] + hidden[:, :, None, :]).expand(-1, -1, 3, -1)
            return obs.sum() + 2.0 * hidden.sum(), {"Xs": Xs}

        def n_step_prediction(self, n, Xs, obs):
            x = Xs.mean(2)[..., :1]
            return [x[:, k:] * (k + 1) for k in range(n + 1)], [obs[:, k:] for k in range(n + 1)]
    FLAGS = Flags(batch_size=2, time=6, MSE_steps=2)
    tr = trainer(StubModel(1, 1), StubSMC(), FLAGS)
    g = np.random.RandomState(5)
    obs, hid = g.randn(7, 6, 1), g.randn(7, 6, 2)           # 4 batches, the last one short
    feed = {tr.obs: obs, tr.hidden: hid}
    z, yh, y, Xs = tr.evaluate(["log_ZSMC", "y_hat", "y", "Xs"], feed)
    n_mine = StubSMC.calls
    want_z = np.array([obs[s:s + 2].sum() + 2 * hid[s:s + 2].sum() for s in range(0, 7, 2)])
    ok = np.allclose(z, want_z, atol=1e-4) and Xs.shape == (7, 6, 3, 2) and len(yh) == 3
    ok = ok and np.allclose(Xs[:, :, 0], obs + hid, atol=1e-5)
    ok = ok and all(np.allclose(y[k], obs[:, k:], atol=1e-6) and yh[k].shape == (7, 6 - k, 1) for k in range(3))
    ok = ok and n_mine == 2                                   # 4 batches over 2 ranks
    avg = tr.evaluate("log_ZSMC", feed, average=True)
    return bool(ok and abs(float(avg) - want_z.mean()) < 1e-4)


def _worker_body(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from psvo_amd import dp
    from psvo_amd.optim import FlatParams
    r, w = dp.init(backend="gloo")
    assert (r, w) == (rank, world) and dp.world_size() == world and dp.rank() == rank
    torch.manual_seed(0)
    fl = dict(Dx=2, Dy=1, n_particles=6, n_particles_for_BSim_proposal=4, use_bootstrap=True, use_2_q=True,
              objective="AESMC", layers=[8])
    B, T = 4, 5
    _, obs = O.fhn_synthetic(B, T, seed=1)
    obs = obs.float()
    noise = O.make_noise(fl, B, T, seed=3, dtype=torch.float32)
    with torch.no_grad():
        _, log = O.OracleAESMC(O.make_params(fl, seed=0, dtype=torch.float32), fl).get_log_ZSMC(obs, noise)
    idx = log["idx_f"]

    # a tiny torch module carrying the oracle's parameters, flattened by FlatParams
    class M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            P = O.make_params(fl, seed=0, dtype=torch.float32)
            self.P = P
            self.ps = torch.nn.ParameterList()
            for k in ("q0", "q1", "q2", "g"):
                for W, b in P[k]["layers"] + [P[k]["mu"]]:
                    self.ps.append(torch.nn.Parameter(W)); self.ps.append(torch.nn.Parameter(b))

        def params(self):
            P, it = {}, iter(self.ps)
            for k in ("q0", "q1", "q2", "g"):
                layers = [(next(it), next(it)) for _ in self.P[k]["layers"]]
                P[k] = {"layers": layers, "mu": (next(it), next(it)), "sigma_raw": self.P[k]["sigma_raw"],
                        "sigma_min": self.P[k]["sigma_min"]}
            return P
    m = M()
    flat = FlatParams(m)
    assert flat.flat.dtype == torch.float32 and flat.numel == sum(p.numel() for p in m.ps)
    dp.broadcast_(flat.flat)

    def elbo(lo, hi):
        nz = {"eps_f": noise["eps_f"][:, :, lo:hi], "idx_f": idx[:, :, lo:hi]}
        z, _ = O.OracleAESMC(m.params(), fl).get_log_ZSMC(obs[lo:hi], nz)
        return z
    # sharded: each rank differentiates the mean over ITS sequences, one all-reduce, divide by world
    lo, hi = dp.shard(B)
    assert (lo, hi) == (rank * 2, rank * 2 + 2)
    flat.zero_grad()
    elbo(lo, hi).backward()
    dp.all_reduce_sum_(flat.grad)
    g_dp = flat.grad.clone() / world
    # reference: the full batch on one process
    flat.zero_grad()
    elbo(0, B).backward()
    g_full = flat.grad.clone()
    ok = torch.allclose(g_dp, g_full, atol=1e-4, rtol=1e-3)
    zs = dp.all_reduce_mean_scalar(elbo(lo, hi).detach())
    ok = ok and abs(float(zs) - float(elbo(0, B))) < 1e-3
    ok = ok and dp.replicas_in_sync(flat.flat)
    if rank == 1:
        flat.flat[0] += 1.0
    ok = ok and not dp.replicas_in_sync(flat.flat)
    try:
        dp.shard(5)
        ok = False
    except ValueError:
        pass
    ok = ok and _sharded_evaluate_ok(rank, world)
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_data_parallel_gradient_allreduce_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = dict(q.get(timeout=240) for _ in procs)
    [p.join(60) for p in procs]
    assert res == {0: True, 1: True}


def test_forced_single_rank_group_runs_the_collectives():
    """PSVO_FORCE_PG=1: the process group exists for ONE rank too, so the collective call path (broadcast, all-reduce, replica
    check) is exercised -- how RCCL's initialisation and the in-step all-reduce are driven on a one-GPU box (bench.py)"""
    import subprocess
    import sys
    code = ("import os, torch; os.environ.update(WORLD_SIZE='1', RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT='29533', "
            "PSVO_FORCE_PG='1');\n"
            "from psvo_amd import dp; import torch.distributed as dist\n"
            "r, w = dp.init(backend='gloo'); assert (r, w) == (0, 1) and dist.is_initialized() and dp._active()\n"
            "x = torch.arange(5.); dp.all_reduce_sum_(x); dp.broadcast_(x); assert x.tolist() == [0., 1., 2., 3., 4.]\n"
            "assert dp.replicas_in_sync(x) and float(dp.all_reduce_mean_scalar(torch.tensor(3.))) == 3.0\n"
            "dist.destroy_process_group(); print('ok')\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "ok" in p.stdout, p.stderr[-2000:]
    # without the switch a single rank creates no group (and pays for no collective)
    code2 = ("import os; os.environ.pop('PSVO_FORCE_PG', None); os.environ.update(WORLD_SIZE='1', RANK='0')\n"
             "from psvo_amd import dp; import torch.distributed as dist\n"
             "dp.init(backend='gloo'); assert not dist.is_initialized() and not dp._active(); print('ok')\n")
    p = subprocess.run([sys.executable, "-c", code2], cwd=root, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "ok" in p.stdout, p.stderr[-2000:]
