"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests).

Replaces MXNet's kvstore('device') push/pull (reference deepim/core/module.py:561-623, :666-688; launch scripts disable GPU P2P:
experiments/deepim/deepim_train_test.py:12) with ONE all-reduce(SUM) of the flat gradient vector per optimizer step -- SUM, not
mean, because the reference sums gradients over GPUs and samples (rescale_grad = 1.0, deepim/train.py:383).
Inference shards the independent (observed, rendered) pairs across ranks with no data-path collective.
"""
import os
import socket
import subprocess
import sys

import torch
import torch.distributed as dist


def _free_port():
    s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks_if_needed(n_ranks, script, argv, extra_env=None):
    """One command drives several GPUs, like the reference's `--gpus 0,1,2,3` (deepim/train.py:425-438 builds one executor per
    context inside one process, DataParallelExecutorGroup.py:303-319).  Here it is one PROCESS per GPU: when `n_ranks` > 1 and this
    process is not already a rank of a torch.distributed.run job (WORLD_SIZE unset), start that job -- `python -m
    torch.distributed.run --nnodes=1 --nproc-per-node n_ranks --master-addr 127.0.0.1 --master-port <free> script argv` -- as a
    CHILD process, relay its output and return its exit code; the caller exits with it.  Returns None when there is nothing to launch.
    Must be called before anything initialises the GPU (a process that touched the GPU must never exec or fork workers on this pool);
    counting devices does not initialise it."""
    if n_ranks <= 1:
        return None
    if "WORLD_SIZE" in os.environ:
        world = int(os.environ["WORLD_SIZE"])
        if world != n_ranks:
            raise SystemExit("{} GPUs / ranks requested but WORLD_SIZE is {}: launch with --nproc-per-node {} (or unset WORLD_SIZE and "
                             "let this script start its own ranks)".format(n_ranks, world, n_ranks))
        return None
    env = dict(os.environ)
    env.update(extra_env or {})
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # the host driver only supports dmabuf IPC (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n_ranks) // n_ranks)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), script] + list(argv)
    sys.stdout.flush()
    return subprocess.call(cmd, env=env)


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


def rank_identity(device_index, rank=0, local_rank=0):
    """what makes this rank's line self-proving: host, device ordinal, marketing name, PCI bus id and uuid of the card it drives"""
    props = torch.cuda.get_device_properties(device_index)
    pci = None
    if hasattr(props, "pci_bus_id"):
        pci = "{:04x}:{:02x}:{:02x}".format(int(getattr(props, "pci_domain_id", 0)), int(props.pci_bus_id), int(getattr(props, "pci_device_id", 0)))
    return {"rank": int(rank), "local_rank": int(local_rank), "host": socket.gethostname(), "device": int(device_index), "name": str(props.name),
            "arch": str(getattr(props, "gcnArchName", "")), "pci_bus_id": pci, "uuid": str(getattr(props, "uuid", "")) or None,
            "cus": int(props.multi_processor_count), "hbm_gb": round(props.total_memory / 2 ** 30, 1), "pid": os.getpid()}


def gather_rank_identities(ident, backend="nccl"):
    """-> [identity of rank 0, rank 1, ...] on every rank.  Under nccl (= RCCL) two ranks on one card are a launch error (they would
    halve each other's throughput and the line would still say n_gpus = N): raise.  gloo rehearsals share a card on purpose."""
    if not is_distributed():
        return [ident]
    parts = [None] * dist.get_world_size()
    dist.all_gather_object(parts, ident)
    check_distinct_devices(parts, backend)
    return parts


def check_distinct_devices(idents, backend="nccl"):
    seen = {}
    for r in idents:
        key = (r["host"], r["uuid"] or r["pci_bus_id"] or r["device"])
        if key in seen and backend == "nccl":
            raise RuntimeError("ranks {} and {} drive the same device {} on {}: one process per GPU".format(seen[key], r["rank"], key[1], key[0]))
        seen.setdefault(key, r["rank"])
    return len(seen)


def gather_floats(value):
    """-> [value on rank 0, rank 1, ...] (python floats) on every rank"""
    if not is_distributed():
        return [float(value)]
    parts = [None] * dist.get_world_size()
    dist.all_gather_object(parts, float(value))
    return parts
