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


# ------------------------------------------------------------------------------- bench.py's own launcher
STUB_RANK = r"""
import json, os, sys, time
rank = int(os.environ["RANK"])
mode = sys.argv[1]
if mode == "ok":
    time.sleep(0.2 * rank)
    if rank == 0:
        print(json.dumps({"metric": "stub", "value": 1.0, "world": int(os.environ["WORLD_SIZE"])}), flush=True)
    sys.exit(0)
if mode == "rank1_dies":            # rank 1 dies at start-up, every other rank waits for it "for ever"
    if rank == 1:
        sys.exit(3)
    if rank == 0:
        print("x" * 300000, flush=True)          # more than a pipe buffer: rank 0 must be drained while it waits
    time.sleep(600)
if mode == "rank0_dies":
    if rank == 0:
        print(json.dumps({"error": "RuntimeError: stub", "rank": 0}), flush=True)
        sys.exit(1)
    time.sleep(600)
"""


@pytest.mark.parametrize("mode,world", [("ok", 4), ("rank1_dies", 4), ("rank0_dies", 2)])
def test_bench_launcher_polls_every_rank(tmp_path, monkeypatch, capsys, mode, world):
    """bench.launch_workers with stub ranks (no GPU, no torch in the children): a clean run relays rank 0's line and
    returns 0; a rank that dies while the others wait (rank 0 blocked with a full pipe, or rank 0 itself dead) makes
    the launcher stop exactly its children after the grace period and print ONE JSON error line with every exit code
    — seconds, not the rendezvous timeout."""
    import json
    import sys
    import time
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
    import bench
    stub = tmp_path / "rank.py"
    stub.write_text(STUB_RANK)
    monkeypatch.setattr(bench, "RANK_GRACE_S", 1.5)
    monkeypatch.setattr(bench, "visible_gpus", lambda: world)
    a = bench.parse(["--gpus", str(world)])
    t0 = time.time()
    rc = bench.launch_workers(a, script=stub, argv=[mode])
    dt = time.time() - t0
    out = capsys.readouterr().out.strip().splitlines()
    assert len(out) == 1, out
    line = json.loads(out[0])
    if mode == "ok":
        assert rc == 0 and line == {"metric": "stub", "value": 1.0, "world": world}
        return
    assert rc != 0 and dt < 30.0, (rc, dt)
    codes = line["rank_exit_codes"]
    assert len(codes) == world and "error" in line and line["n_gpus"] == world
    if mode == "rank1_dies":
        assert codes[1] == 3 and all(c not in (0, None) for c in codes) and "rank 1" in line["error"]
    else:
        assert codes[0] == 1 and line["rank0_error"] == "RuntimeError: stub"
