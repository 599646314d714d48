"""Two data-parallel ranks through the real HIP path (both on the one card of the test box, gloo for the collectives: RCCL
refuses two ranks per device): trainer.train shards every mini-batch over the ranks, all-reduces the flat gradient, applies
the identical Adam update, checks the replicas bit for bit every print_freq epochs and shards the evaluation batches."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, tmp, q):
    try:
        os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                          HSA_ENABLE_IPC_MODE_LEGACY="0")
        sys.path.insert(0, ROOT)
        from oracle import psvo_oracle as O
        from psvo_amd import dp
        from psvo_amd.model import SSM
        from psvo_amd.SMC.PSVO import PSVO
        from psvo_amd.trainer import trainer
        from tests import helpers as Hh
        torch.cuda.set_device(0)
        dp.init(backend="gloo")
        os.chdir(tmp)
        hid, obs = O.fhn_synthetic(12, 24, seed=0)
        FLAGS = Hh.make_flags("PSVO", n_particles=16, n_particles_for_BSim_proposal=4, batch_size=4, time=24, epoch=3,
                              lr=1e-2, MSE_steps=3, saving_num=4, rslt_dir_name="t")
        torch.manual_seed(rank)                       # different initial replicas: train() broadcasts rank 0's
        np.random.seed(0)                             # (the shuffle must be the same on every rank)
        model = SSM(FLAGS).cuda()
        smc = PSVO(model, FLAGS)
        smc.generator = torch.Generator(device="cuda").manual_seed(100 + rank)
        tr = trainer(model, smc, FLAGS)
        rlt = os.path.join(tmp, "rslts", "t", "run") + "/"
        if rank == 0:
            os.makedirs(rlt, exist_ok=True)
        torch.distributed.barrier()
        tr.init_data_saving(rlt)
        hist, _ = tr.train(obs[:8].numpy(), obs[8:].numpy(), hid[:8].numpy(), hid[8:].numpy(), print_freq=1)
        flat = tr.flat.flat.detach().cpu()
        ok = dp.replicas_in_sync(tr.flat.flat)
        q.put((rank, ok, float(flat.double().sum()), hist["log_ZSMC_trains"]))
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    except Exception as e:   # report instead of letting the parent wait for its timeout
        import traceback
        traceback.print_exc()
        q.put((rank, False, repr(e), []))


@pytest.mark.timeout(600)
def test_two_rank_training_keeps_replicas_identical(built_lib, tmp_path):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=500) for _ in procs)
    [p.join(60) for p in procs]
    (r0, ok0, sum0, h0), (r1, ok1, sum1, h1) = res
    assert ok0 is True and ok1 is True, res
    assert sum0 == sum1                                   # bit-identical parameters on both ranks
    assert h0 == h1 and len(h0) == 4 and h0[-1] > h0[0]   # same (gathered) metrics, ELBO went up
    assert os.path.exists(os.path.join(str(tmp_path), "rslts", "epoch_data", "t", "run", "metric_3.p"))
