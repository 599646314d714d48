"""Data-parallel sharding of the batch of sequences: one process per GPU (torch.distributed,
backend "nccl" = RCCL over xGMI on ROCm; "gloo" in CPU tests).

The reference has no distributed code at all; the path shards naturally
over sequences because no op couples different batch elements before the final mean
(reference src/SMC/SVO.py:309, src/SMC/PSVO.py:65).  Per training step there is exactly ONE
collective: an all-reduce(SUM) of the flat gradient buffer (<100 KB, latency-bound), after which
every rank applies the identical Adam update, so replicas stay bit-identical.
"""
import os

import torch
import torch.distributed as dist


def init(backend=None, device=None):
    """Initialise the default process group from the torchrun environment (RANK, WORLD_SIZE,
    MASTER_ADDR, MASTER_PORT).  Returns (rank, world_size).  A single process needs no group."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    # PSVO_FORCE_PG=1: create the group (and issue the collectives) even for one rank -- the only way to drive RCCL's
    # initialisation and the all-reduce call path on a one-GPU box
    if (world > 1 or _forced()) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world


def _forced():
    return os.environ.get("PSVO_FORCE_PG", "0") == "1"


def _active():
    return dist.is_initialized() and (dist.get_world_size() > 1 or _forced())


def world_size():
    return dist.get_world_size() if dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_initialized() else 0


def shard(n_items, r=None, w=None):
    """[lo, hi) of the equal contiguous shard of rank r; n_items % w == 0."""
    r = rank() if r is None else r
    w = world_size() if w is None else w
    if n_items % w != 0:
        raise ValueError("batch of %d sequences does not split evenly over %d ranks" % (n_items, w))
    k = n_items // w
    return r * k, (r + 1) * k


def all_reduce_sum_(flat):
    """The single gradient collective of a training step (in place)."""
    if _active():
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return flat


def all_reduce_mean_scalar(x):
    """Mean over ranks of a scalar tensor (validation / logged ELBO)."""
    if _active():
        x = x.detach().clone()
        dist.all_reduce(x, op=dist.ReduceOp.SUM)
        x = x / dist.get_world_size()
    return x


def broadcast_(flat, src=0):
    """Make every replica start from rank `src`'s parameters."""
    if _active():
        dist.broadcast(flat, src=src)
    return flat


def replicas_in_sync(flat):
    """Cheap periodic check that replicas are bit-identical: max |theta - theta_rank0| == 0."""
    if not _active():
        return True
    ref = flat.detach().clone()
    dist.broadcast(ref, src=0)
    d = (ref - flat).abs().max()
    dist.all_reduce(d, op=dist.ReduceOp.MAX)
    return float(d) == 0.0
