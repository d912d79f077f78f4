"""-m gpu: BASELINE config 5 on one device — KukaButtonGymEnv raw_pixels 64x64, 32 768 envs as 8 shards of 4096 (global env
ids g*4096 ..) against ONE 32 768-env handle (SURVEY 7 test plan "sharding invariance: N = 32 768 on 1 vs 8 GPUs (or
simulated shards on 1)"; SURVEY 8e: seeds are seed0 + global id, so trajectories must not depend on the sharding).

Frames, encoder states, rewards, dones and the device-side episode statistics must be bit-equal; the path's one collective
(all-gather of per-env episode returns) is run over 8 simulated ranks with gloo on CPU tensors and must reproduce the single
handle's plane.  Both sides run the full-model tree lane-group kernel (it steps every batch size), so bit-equality is meaningful."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from srlhip import _lib
from srlhip.pixel_env import PixelStateVecEnv
from state_representation.models import SRLNeuralNetwork

pytestmark = pytest.mark.gpu
G, PER, T = 8, 4096, 64


def _gather_worker(rank, world, port, shards, out):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from srlhip import sharding
    got = sharding.gather_episode_returns(shards[rank].clone())
    if rank == 0:
        out.copy_(got)
    dist.barrier()
    dist.destroy_process_group()


def test_eight_shards_equal_one_handle():
    torch.manual_seed(0)
    dev = torch.device("cuda", 0)
    enc = SRLNeuralNetwork(3, cuda=True, img_shape=(64, 64), device=dev)
    assert enc.backend == "hip"
    kw = {"random_target": True}
    # the arm needs hundreds of steps to reach anything (0.35 rad/s joint speed limit), so the step counters are advanced to
    # just below the 1001-step limit, staggered by global env id: every env ends an episode (and auto-resets) inside the window
    counters = (1001 - 10 - (np.arange(G * PER) % 40)).astype(np.int32)
    full = PixelStateVecEnv("KukaButtonGymEnv-v0", G * PER, enc, seed=0, img_shape=(64, 64), env_kwargs=kw)
    assert full.h.kuka_kernel() == "tree"
    rs = np.random.RandomState(7)
    actions = torch.from_numpy(rs.randint(6, size=(T, G * PER)).astype(np.int32)).to(dev)
    st0 = full.reset().clone()
    img0 = full.images.clone()
    full.h.set_state(_lib.F_STEP_COUNT, counters)
    keep = [10, T - 1]                                        # full planes kept at two steps; checksums at every step
    ref = {"states": [], "rew": [], "done": [], "img_sum": [], "img": {}}
    for t in range(T):
        s, r, d = full.step(actions[t])
        torch.cuda.synchronize()
        ref["states"].append(s.clone()); ref["rew"].append(r.clone()); ref["done"].append(d.clone())
        ref["img_sum"].append(full.images.view(G, -1).to(torch.int64).sum(1).cpu())
        if t in keep:
            ref["img"][t] = full.images.clone()
    ret_full = torch.zeros(G * PER, dtype=torch.float32, device=dev)
    len_full = torch.zeros(G * PER, dtype=torch.int32, device=dev)
    fin_full = torch.zeros(G * PER, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()                                   # the zero fills run on torch's stream, the statistics kernel on the stepper's
    full.h.episode_stats_device(last_return=ret_full.data_ptr(), last_length=len_full.data_ptr(), n_finished=fin_full.data_ptr())
    full.h.sync()
    full.close()
    assert int(torch.stack(ref["done"]).sum()) == G * PER     # every env crossed the step limit once

    shard_returns = []
    for g in range(G):
        sl = slice(g * PER, (g + 1) * PER)
        env = PixelStateVecEnv("KukaButtonGymEnv-v0", PER, enc, seed=0, img_shape=(64, 64), first_env_id=g * PER, env_kwargs=kw)
        s = env.reset()
        torch.cuda.synchronize()
        assert torch.equal(s, st0[sl]) and torch.equal(env.images, img0[sl])
        env.h.set_state(_lib.F_STEP_COUNT, counters[sl])
        for t in range(T):
            a = actions[t, sl].contiguous()
            s, r, d = env.step(a)
            torch.cuda.synchronize()
            assert torch.equal(r, ref["rew"][t][sl]) and torch.equal(d, ref["done"][t][sl]), (g, t)
            assert torch.equal(s, ref["states"][t][sl]), (g, t)
            assert int(env.images.to(torch.int64).sum()) == int(ref["img_sum"][t][g]), (g, t)
            if t in keep:
                assert torch.equal(env.images, ref["img"][t][sl]), (g, t)
        ret = torch.zeros(PER, dtype=torch.float32, device=dev)
        ln = torch.zeros(PER, dtype=torch.int32, device=dev)
        fin = torch.zeros(PER, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        env.h.episode_stats_device(last_return=ret.data_ptr(), last_length=ln.data_ptr(), n_finished=fin.data_ptr())
        env.h.sync()
        assert torch.equal(ret, ret_full[sl]) and torch.equal(ln, len_full[sl]) and torch.equal(fin, fin_full[sl])
        shard_returns.append(ret.cpu())
        env.close()

    # the one collective of the path over 8 simulated ranks (gloo, CPU tensors): rank-major == global env id order
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    out = torch.zeros(G * PER, dtype=torch.float32).share_memory_()
    shards = [s.share_memory_() for s in shard_returns]
    mp.spawn(_gather_worker, args=(G, port, shards, out), nprocs=G, join=True)
    assert torch.equal(out, ret_full.cpu())
