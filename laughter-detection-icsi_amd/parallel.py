"""Data-parallel training over the GPUs of one node: one process per GPU, RCCL over xGMI.

The reference is single-GPU (cluster_scripts/train_laugh_job.sh:32) and has no collective; the path shards over
segments, so the only exchange is ONE all-reduce of the flat fp32 gradient buffer
(221k floats, < 1 MB) per step.  The 1/world_size of the mean is folded into the clip+Adam kernel (grad_scale), so
the collective is a plain sum.  BatchNorm statistics stay local to each rank (standard DDP semantics); running
statistics of rank 0 are the ones checkpointed.

Backend: "nccl" (= RCCL on ROCm) on GPUs; "gloo" for the CPU tests of the sharding/averaging logic.

Launching.  Ranks come either from an outer launcher (`python -m torch.distributed.run ...` exports RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_*) or from `spawn_ranks`, which bench.py / train.py / segment_laughter.py call when `--gpus N > 1`
is asked for and no launcher environment is present: the parent starts N fresh children and only waits for them.  The
parent never touches the GPU -- devices are counted from the KFD topology in sysfs, not through the HIP runtime -- and
never re-execs: a process that has initialised the GPU must not be replaced by another program.
"""
import os
import socket
import subprocess
import sys
import time

import torch
import torch.distributed as dist


def under_launcher():
    return "RANK" in os.environ and "WORLD_SIZE" in os.environ


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def visible_gpu_count():
    """GPUs this process would see, WITHOUT initialising the HIP / HSA runtime in it: KFD topology nodes with SIMDs
    (/sys/class/kfd/kfd/topology/nodes/*/properties, `simd_count` > 0; CPUs are nodes with 0), cut down by
    HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES when set.  Falls back to torch.cuda.device_count()
    (which may load the runtime on builds without amdsmi) only when sysfs is not readable."""
    root = "/sys/class/kfd/kfd/topology/nodes"
    n = None
    try:
        n = 0
        for node in os.listdir(root):
            try:
                props = dict(line.split(None, 1) for line in open(os.path.join(root, node, "properties")) if " " in line)
            except OSError:
                continue
            if int(props.get("simd_count", "0").strip() or 0) > 0:
                n += 1
    except OSError:
        n = None
    if n is None or n == 0:
        # (no KFD nodes visible: a container without sysfs access, or no GPU -- let torch say which)
        return torch.cuda.device_count()
    try:   # a container may see the host's whole topology in sysfs but only some render nodes in /dev/dri
        nodes = [d for d in os.listdir("/dev/dri") if d.startswith("renderD")]
        if nodes:
            n = min(n, len(nodes))
    except OSError:
        pass
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            ids = [x for x in v.split(",") if x.strip() != ""]
            n = min(n, len(ids))
    return n


SPAWN_TIMEOUT_S = 3600.0   # what bench.py passes for its self-launched ranks; train.py / segment_laughter.py run unlimited
GRACE_S = 10.0             # between terminate() and kill() of ranks that outlive a failed peer or the limit


def spawn_ranks(n_ranks, script, argv, need_gpus=True, extra_env=None, timeout=None):
    """Start `n_ranks` children `python script *argv`, one per GPU, with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR /
    MASTER_PORT set; wait for all of them; return the worst exit code.

    Rank 0 inherits stdout (its single JSON line / log is the job's output); the other ranks' stdout goes to stderr.
    If fewer than `n_ranks` devices are visible the job is refused (exit code 2) instead of running on fewer GPUs under
    an N-GPU label.  `timeout` = None (the default: training runs for hours) never stops a healthy job; bench.py and the
    tests pass a finite limit.  When one child fails, or the job outlives `timeout` seconds, the others are terminated by PID (a
    rank waiting in a collective for a dead peer would otherwise hang until the RCCL timeout) and, if they ignore that for
    GRACE_S seconds (a rank stuck inside a HIP call does), killed.  The ranks meet through a file store in a private
    temporary directory, not through a TCP port picked here (which could be taken before rank 0 binds it)."""
    n_ranks = int(n_ranks)
    if need_gpus and os.environ.get("LAD_REHEARSE_ON_ONE_GPU") == "1":
        need_gpus = False   # rehearsal (tests): all ranks on device 0 over gloo, see init_from_env
    if need_gpus:
        have = visible_gpu_count()
        if have < n_ranks:
            sys.stderr.write(f"{os.path.basename(script)}: {n_ranks} GPUs requested but {have} visible: refusing to run "
                             f"(a {n_ranks}-GPU figure measured on fewer devices would be mislabelled)\n")
            return 2
    import shutil
    import tempfile
    rdzv_dir = tempfile.mkdtemp(prefix="lad_rdzv_")
    env = dict(os.environ)
    # Rendezvous through a FILE store (init_from_env honours LAD_RDZV_FILE): a TCP port chosen here could be taken by
    # somebody else before rank 0 binds it.  MASTER_ADDR / MASTER_PORT are still exported for code that reads them.
    env.update({"WORLD_SIZE": str(n_ranks), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(_free_port()),
                "LAD_RDZV_FILE": os.path.join(rdzv_dir, "store"),
                "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
                "LAD_SPAWNED": "1"})
    env.update(extra_env or {})
    procs = []
    for r in range(n_ranks):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, script] + list(argv), env=e,
                                      stdout=None if r == 0 else sys.stderr))
    t0 = time.time()
    worst = 0
    alive = set(range(n_ranks))
    kill_at = None

    def stop_all():
        nonlocal kill_at
        for o in alive:
            procs[o].terminate()
        if kill_at is None:
            kill_at = time.time() + GRACE_S

    try:
        while alive:
            for r in sorted(alive):
                rc = procs[r].poll()
                if rc is None:
                    continue
                alive.discard(r)
                if rc != 0:
                    worst = rc if worst == 0 else worst
                    if alive:
                        sys.stderr.write(f"rank {r} exited with code {rc}; stopping the other ranks\n")
                        stop_all()
            if alive:
                now = time.time()
                if timeout is not None and kill_at is None and now - t0 > timeout:
                    sys.stderr.write(f"ranks {sorted(alive)} still running after {timeout} s: terminating\n")
                    stop_all()
                    worst = worst or 124
                if kill_at is not None and now > kill_at:
                    sys.stderr.write(f"ranks {sorted(alive)} ignored SIGTERM for {GRACE_S} s: killing\n")
                    for o in alive:
                        procs[o].kill()
                    kill_at = now + 3600.0   # (killed processes are reaped by the polls above)
                time.sleep(0.05)
    finally:
        shutil.rmtree(rdzv_dir, ignore_errors=True)
    return worst if worst >= 0 else 128 - worst  # a child killed by signal s reports -s


def init_from_env(backend=None):
    """Initialise torch.distributed from torchrun's environment; returns (rank, world, local_rank).
    A single process without RANK/WORLD_SIZE gives (0, 1, 0) and no process group."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("LAD_REHEARSE_ON_ONE_GPU") == "1":
        # Rehearsal of the N-rank code path on a one-GPU box (tests/test_bench_gpu.py): every rank uses device 0 and the
        # collectives go through gloo (RCCL refuses two ranks on one device).  Never a measurement: bench.py labels the
        # line with the backend it ran on.
        local, backend = 0, "gloo"
    launched = under_launcher()  # torchrun / spawn_ranks: also initialise a 1-rank group
    if (world > 1 or launched) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if os.environ.get("LAD_RDZV_FILE"):   # spawn_ranks: file-store rendezvous (no port to lose a race for)
            kw["init_method"] = "file://" + os.environ["LAD_RDZV_FILE"]
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local), **kw)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, **kw)
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


def train_step_count(n_items, batch, world, min_batch=2):
    """Optimiser steps per epoch -- THE SAME NUMBER ON EVERY RANK, computed from n_items alone.

    Segments are dealt round-robin (rank r owns positions r, r+world, ... of the index order), so shard sizes differ by
    at most one; step s takes the next `batch` positions of each shard.  Every step issues one all-reduce, so a rank
    with one step more or less than its peers would pair collectives across epochs and finally hang.  The ragged last
    step is kept only if EVERY rank still has at least `min_batch` segments in it (train-mode BatchNorm needs two;
    the reference's loop has the same limit, torch raises on a batch of one): the decision is global, never per rank."""
    if n_items <= 0 or batch < min_batch:
        return 0
    longest = (n_items + world - 1) // world      # rank 0
    shortest = n_items // world                   # rank world-1
    steps = (longest + batch - 1) // batch
    if shortest - (steps - 1) * batch < min_batch:
        steps -= 1
    return max(steps, 0)


def train_batches(n_items, batch, rank, world, min_batch=2):
    """Index arrays (positions in the epoch's segment order) of this rank's batches: train_step_count() of them."""
    import numpy as np
    mine = np.arange(rank, n_items, world, dtype=np.int64)
    for s in range(train_step_count(n_items, batch, world, min_batch)):
        yield mine[s * batch:(s + 1) * batch]


class GradReducer:
    """Sum-all-reduce of the flat gradient buffer; pass as `grad_reduce=` to ResNetBigger.train_step together with
    `grad_scale=reducer.scale` (mean over ranks, applied inside the clip+Adam kernel)."""

    def __init__(self, group=None):
        self.group = group
        self.active = dist.is_initialized()
        self.world = dist.get_world_size(group) if self.active else 1
        self.backend = dist.get_backend(group) if self.active else None
        self.scale = 1.0 / self.world
        self.calls = 0          # collectives issued (tests / bench.py: proof that the gradient went through all_reduce)
        self.events = None      # bench.py: list that receives (start, end) HIP events around every collective

    def __call__(self, flat_grad):
        if self.active:  # also with one rank under a launcher: same code path as N > 1
            timed = self.events is not None and flat_grad.is_cuda
            if timed:
                e0 = torch.cuda.Event(enable_timing=True)
                e0.record()
            dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=self.group)
            if timed:  # the launch stream waits for the collective, so this event fires after it
                e1 = torch.cuda.Event(enable_timing=True)
                e1.record()
                self.events.append((e0, e1))
            self.calls += 1
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
    """Counter vectors of the head -- (8,) or (steps, 8): [mean BCE, #correct, #pred+, #true+, #target+, n, ..] -- summed over
    the ranks: row s becomes the counter vector of the GLOBAL batch of step s (the union of the ranks' batches; the loss
    entry is the mean over its segments, i.e. the rank losses weighted by their batch sizes).  Logging cadence only
    (all-reduce 4 counters + loss sum); every rank gets the result."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return metrics
    m = metrics.clone()
    m[..., 0] *= m[..., 5]                     # loss sum of the rank's batch
    dist.all_reduce(m, op=dist.ReduceOp.SUM)
    m[..., 0] /= m[..., 5].clamp_min(1.0)
    return m


def sum_rows(mat):
    """Sum-all-reduce of a matrix whose rows are owned by different ranks (zeros elsewhere): sharded validation."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return mat
    m = mat.clone()
    dist.all_reduce(m, op=dist.ReduceOp.SUM)
    return m


def collective_device():
    """Device of the tensors this process hands to collectives: its GPU when it has one (nccl, or gloo staging through the
    host in the one-GPU rehearsal), the CPU otherwise (gloo tests)."""
    return torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")


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
