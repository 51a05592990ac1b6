"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests).

Replaces MXNet's kvstore('device') push/pull (reference deepim/core/module.py:561-623, :666-688; launch scripts disable GPU P2P:
experiments/deepim/deepim_train_test.py:12) with ONE all-reduce(SUM) of the flat gradient vector per optimizer step -- SUM, not
mean, because the reference sums gradients over GPUs and samples (rescale_grad = 1.0, deepim/train.py:383).
Inference shards the independent (observed, rendered) pairs across ranks with no data-path collective.
"""
import torch
import torch.distributed as dist


def is_distributed():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def shard_range(total, rank, world):
    """contiguous slice [begin, end) of `total` independent pairs owned by `rank` (sizes differ by at most one)"""
    base, rem = divmod(total, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def even_shard_range(total, rank, world):
    """contiguous slice [begin, end) with the SAME length floor(total / world) on every rank; the remainder is dropped.
    Training needs this: each optimizer step posts a blocking all-reduce, so a rank with one batch more than its peers would wait
    for a collective nobody else issues."""
    per = total // world
    return rank * per, rank * per + per


def allreduce_sum_(flat, group=None):
    """in-place SUM over ranks of the flat gradient bucket (no-op in a single process)"""
    if is_distributed():
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return flat


def max_over_ranks(value, device="cpu"):
    """MAX over ranks of a python float (bench timing)"""
    if not is_distributed():
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if is_distributed():
        dist.barrier()
