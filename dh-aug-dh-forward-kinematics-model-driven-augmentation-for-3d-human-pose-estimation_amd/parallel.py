"""Data-parallel helpers: one process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on ROCm, "gloo" in the
CPU tests).  The augmentation batch shards across ranks; weights are replicated; the only exchange is the flat
gradient bucket of the network being stepped (optim.FusedAdam.exchange)."""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torch.distributed.run).
    Returns (rank, world, local_rank); a no-op for single-process runs."""
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend or ("nccl" if torch.cuda.is_available() else "gloo"))
    if torch.cuda.is_available():
        torch.cuda.set_device(local)
    return rank, world, local


def shard_range(total, rank, world):
    """[begin, end) of this rank's contiguous shard of `total` units (videos stay whole: shard B, not B*R)."""
    base, rem = divmod(total, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def rank_seed(seed, rank):
    """per-rank RNG stream for noise / jitter / GP alpha / camera draws"""
    return int(seed) * 1000003 + int(rank)


def broadcast_parameters(modules, src=0):
    """replicas start from rank `src`'s weights"""
    if not (dist.is_available() and dist.is_initialized()):
        return
    for m in modules:
        for p in m.parameters():
            dist.broadcast(p.data, src)
    # the write went through .data (no version bump): every packed bf16 / fragment copy of a weight is stale now
    from . import autograd_ops as A
    A.bump_weight_epoch()


def broadcast_optimizers(optimizers, src=0):
    """same, one collective per network: FusedAdam's flat parameter buffer (the parameters are views of it)"""
    if not (dist.is_available() and dist.is_initialized()):
        return
    for o in optimizers:
        dist.broadcast(o.flat_param, src)
    from . import autograd_ops as A
    A.bump_weight_epoch()
