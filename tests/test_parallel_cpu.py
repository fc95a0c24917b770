"""World-size-2 gloo tests (CPU) of the data-parallel helpers: sharding, gradient averaging, probability gather."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "laughter-detection-icsi_amd"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import parallel
    r, w, _ = parallel.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    # 1. gradient averaging: sum all-reduce + the 1/world scale equals the gradient of the concatenated batch
    torch.manual_seed(0)
    wgt = torch.randn(7, 3)
    data = torch.randn(8, 7)              # global batch of 8, sharded contiguously
    idx = list(parallel.shard_indices(8, rank, world))
    local = data[idx]
    g_local = (local @ wgt).sum(0) / len(idx)  # stand-in for a per-rank mean gradient
    red = parallel.GradReducer()
    g = red(g_local.clone()) * red.scale
    g_ref = (data @ wgt).sum(0) / 8
    ok_grad = torch.allclose(g, g_ref, atol=1e-6)
    # 2. counters
    m = torch.tensor([0.5 + rank, 3.0, 1.0, 1.0, 2.0, 4.0, 0.0, 0.0])
    mr = parallel.reduce_counters(m)
    ok_cnt = abs(float(mr[0]) - 1.0) < 1e-6 and float(mr[1]) == 6.0 and float(mr[5]) == 8.0
    # 3. inference gather with a ragged tail (11 windows over 2 ranks -> 6 + 5)
    n = 11
    mine = torch.tensor([float(i) for i in parallel.shard_indices(n, rank, world)])
    allp = parallel.gather_probs(mine, n, rank, world)
    ok_gather = torch.equal(allp, torch.arange(n, dtype=torch.float32))
    # 4. parameter broadcast
    lin = torch.nn.Linear(3, 2)
    with torch.no_grad():
        lin.weight.fill_(float(rank + 1))
    parallel.broadcast_parameters(lin, src=0)
    ok_bc = float(lin.weight.max()) == 1.0 and float(lin.weight.min()) == 1.0
    q.put((rank, ok_grad, ok_cnt, ok_gather, ok_bc))
    dist.destroy_process_group()


def test_world_size_two_gloo():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for r in res:
        assert all(r[1:]), r


def test_shard_indices_cover_disjointly():
    import parallel
    for n in (0, 1, 7, 8, 360000, 360001):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                seen.extend(parallel.shard_indices(n, r, world))
            assert seen == list(range(n))


def test_single_process_is_a_noop():
    import parallel
    red = parallel.GradReducer()
    assert red.world == 1 and red.scale == 1.0
    g = torch.ones(4)
    assert red(g) is g
    assert parallel.shard_indices(10, 0, 1) == range(0, 10)
