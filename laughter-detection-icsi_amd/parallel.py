"""Data-parallel training over the GPUs of one node: one process per GPU, RCCL over xGMI.

The reference is single-GPU (cluster_scripts/train_laugh_job.sh:32) and has no collective; the path shards over
segments, so the only exchange is ONE all-reduce of the flat fp32 gradient buffer
(221k floats, < 1 MB) per step.  The 1/world_size of the mean is folded into the clip+Adam kernel (grad_scale), so
the collective is a plain sum.  BatchNorm statistics stay local to each rank (standard DDP semantics); running
statistics of rank 0 are the ones checkpointed.

Backend: "nccl" (= RCCL on ROCm) on GPUs; "gloo" for the CPU tests of the sharding/averaging logic.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from torchrun's environment; returns (rank, world, local_rank).
    A single process without RANK/WORLD_SIZE gives (0, 1, 0) and no process group."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    under_launcher = "RANK" in os.environ and "WORLD_SIZE" in os.environ  # torchrun: also initialise a 1-rank group
    if (world > 1 or under_launcher) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


def world_size():
    return dist.get_world_size() if dist.is_initialized() else 1


def shard_indices(n_items, rank, world):
    """Segment indices owned by `rank`: the contiguous block [rank*ceil(n/world), ...) clipped to n.
    Blocks are disjoint, ordered and cover range(n) (tests/test_parallel_cpu.py)."""
    per = (n_items + world - 1) // world
    lo = min(n_items, rank * per)
    hi = min(n_items, lo + per)
    return range(lo, hi)


class GradReducer:
    """Sum-all-reduce of the flat gradient buffer; pass as `grad_reduce=` to ResNetBigger.train_step together with
    `grad_scale=reducer.scale` (mean over ranks, applied inside the clip+Adam kernel)."""

    def __init__(self, group=None):
        self.group = group
        self.active = dist.is_initialized()
        self.world = dist.get_world_size(group) if self.active else 1
        self.scale = 1.0 / self.world

    def __call__(self, flat_grad):
        if self.active:  # also with one rank under a launcher: same code path as N > 1
            dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=self.group)
        return flat_grad


def broadcast_parameters(model, src=0):
    """Make every rank start from rank `src`'s parameters and running statistics."""
    if not dist.is_initialized():
        return
    eng = getattr(model, "engine", None)
    if eng is not None and next(model.parameters()).is_cuda:
        dist.broadcast(eng.flat_param(), src)
        eng.notify_weights_changed()
    else:
        for p in model.parameters():
            dist.broadcast(p.data, src)
    for b in model.buffers():
        dist.broadcast(b, src)


def reduce_counters(metrics):
    """Sum the head's counter vector over ranks (loss entry becomes the mean): logging cadence only."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return metrics
    m = metrics.clone()
    dist.all_reduce(m, op=dist.ReduceOp.SUM)
    m[0] /= dist.get_world_size()
    return m


def gather_probs(local_probs, n_total, rank, world):
    """Inference: concatenate per-rank probability shards (contiguous shard_indices order) on every rank."""
    if world == 1:
        return local_probs
    per = (n_total + world - 1) // world
    pad = torch.zeros(per, device=local_probs.device, dtype=local_probs.dtype)
    pad[:local_probs.numel()] = local_probs
    out = [torch.zeros_like(pad) for _ in range(world)]
    dist.all_gather(out, pad)
    return torch.cat(out)[:n_total]
