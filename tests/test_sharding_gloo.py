"""world_size-2 `gloo` test of the N>1 path: shard index math, global-id seeding and the
episode-return all-gather.  The env stepping inside each rank is done by the CPU oracle here
(no GPU in this suite); the GPU twin is tests/test_gpu_mobile.py::test_sharding_invariance."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from oracle import clib
from srlhip import sharding

N_TOTAL, T, SEED0 = 64, 300, 17


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, actions, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    sharding.init_process_group("gloo")
    first, count = sharding.shard_range(N_TOTAL, world, rank)
    assert sharding.dist_env() == (rank, rank, world)
    ora = clib.mobile_rollout(0, SEED0 + first + np.arange(count), T, actions=np.ascontiguousarray(actions[:, first:first + count]),
                              random_target=True)
    local = torch.from_numpy(ora["reward"].sum(axis=0).astype(np.float32))
    gathered = sharding.gather_episode_returns(local)
    slowest = sharding.max_over_ranks(1.0 + rank)
    q.put((rank, gathered.numpy().copy(), slowest))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_two_rank_sharding_and_return_allgather():
    clib.build()
    actions = np.random.RandomState(0).randint(4, size=(T, N_TOTAL)).astype(np.int32)
    full = clib.mobile_rollout(0, SEED0 + np.arange(N_TOTAL), T, actions=actions, random_target=True)
    expect = full["reward"].sum(axis=0).astype(np.float32)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, actions, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, gathered, slowest in results:
        assert np.array_equal(gathered, expect), rank       # identical on every rank, global env id order
        assert slowest == 2.0


def test_shard_range():
    assert [sharding.shard_range(32768, 8, r) for r in (0, 3, 7)] == [(0, 4096), (12288, 4096), (28672, 4096)]
    with pytest.raises(ValueError):
        sharding.shard_range(10, 4, 0)
    t = torch.arange(4, dtype=torch.float32)
    assert sharding.gather_episode_returns(t) is t          # single process: no collective
