"""Batch sharding across the GPUs of one node.

QP instances of a batch are independent (no coupling term anywhere in the reference's
solve loop, reluqpth.py:201-249), so the path shards embarrassingly: one process per GPU,
rank r owns a contiguous slice of the instances, and there is NO collective on the data
path.  ``torch.distributed`` (backend "nccl" = RCCL on ROCm, "gloo" on CPU tests) is used
only to bracket timed regions with a barrier and to reduce the scalars that are reported
(max elapsed time, total iterations, total solved).
"""
import os

import torch


def shard_range(total, rank, world):
    """Contiguous split of ``total`` instances: returns (start, size) of ``rank``'s shard.
    Sizes differ by at most one; the union over ranks is exactly range(total)."""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world of %d" % (rank, world))
    base, rem = divmod(total, world)
    start = rank * base + min(rank, rem)
    return start, base + (1 if rank < rem else 0)


def env_rank():
    """(rank, world_size, local_rank) from the torchrun environment (defaults: single process)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def local_device(local_rank):
    """The GPU of this rank: LOCAL_RANK, folded into the visible device count (a launcher that narrows
    HIP_VISIBLE_DEVICES to one GPU per process leaves every rank with device 0)."""
    nd = torch.cuda.device_count()
    return torch.device("cuda", local_rank % nd if nd > 0 else 0)


def init(backend=None):
    """Join the process group when WORLD_SIZE > 1.  Returns (rank, world, local_rank, dist|None)."""
    rank, world, local_rank = env_rank()
    if world <= 1:
        return rank, world, local_rank, None
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend is None:      # RQP_DIST_BACKEND=gloo: rehearse several ranks on ONE GPU (RCCL refuses two ranks per device)
        backend = os.environ.get("RQP_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if not dist.is_initialized():
        if backend == "nccl":
            dev = local_device(local_rank)
            torch.cuda.set_device(dev)
            dist.init_process_group(backend, device_id=dev)
        else:
            dist.init_process_group(backend)
    return rank, world, local_rank, dist


def reduce_report(dist, device, elapsed_s, sum_iters, n_solved, n_qps, extra_max=()):
    """Whole-job numbers: MAX over ranks of the times, SUM over ranks of the counts.
    Returns (elapsed_max, [extra maxima...], total_iters, total_solved, total_qps)."""
    if dist is not None and dist.get_backend() == "gloo":
        device = "cpu"
    tmax = torch.tensor([elapsed_s] + list(extra_max), dtype=torch.float64, device=device)
    tsum = torch.tensor([float(sum_iters), float(n_solved), float(n_qps)], dtype=torch.float64, device=device)
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
    tm = [float(v) for v in tmax.cpu()]
    ts = [float(v) for v in tsum.cpu()]
    return tm[0], tm[1:], ts[0], ts[1], ts[2]


def gather_shards(dist, local, total, rank, world):
    """Optional convenience (not on the timed path): all-gather per-instance results of unequal
    shard sizes into one tensor of ``total`` rows on every rank."""
    if dist is None:
        return local
    sizes = [shard_range(total, r, world)[1] for r in range(world)]
    mx = max(sizes)
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    outs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(outs, pad)
    return torch.cat([o[:s] for o, s in zip(outs, sizes)], dim=0)
