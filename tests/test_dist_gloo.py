"""N>1 path on CPU: gloo processes (world 2, and world 8 = the node the scaling series runs on) shard an ensemble
by member and collect the trajectories with the one all-gather the design uses (rollout.gather_trajectories);
even (64 members over 8 ranks: BASELINE configs[2]) and uneven (61 over 8, 5 over 2) member counts."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from molecular_dynamics_neural_operator_amd.rollout import gather_trajectories, shard_members
        ids = shard_members(total, rank, world)
        T, N = 4, 5
        # member m's "trajectory" is filled with m + 0.01*t so the global order is checkable
        local = torch.stack([torch.full((T, N, 3), float(m)) + 0.01 * torch.arange(T).view(T, 1, 1) for m in ids], 1) \
            if ids else torch.zeros((T, 0, N, 3))
        full = gather_trajectories(local, total)
        ok = full.shape == (T, total, N, 3)
        for m in range(total):
            ok = ok and torch.allclose(full[:, m, 0, 0], m + 0.01 * torch.arange(T, dtype=torch.float32))
        # max-over-ranks timing reduction as bench.py does it
        tmax = torch.tensor([float(rank + 1)])
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        ok = ok and float(tmax) == float(world)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,total", [(2, 8), (2, 5), (8, 64), (8, 61), (8, 5)])
def test_member_sharding_and_allgather(world, total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert res == [(r, True) for r in range(world)]
