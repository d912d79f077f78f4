"""-m gpu: the EXACT configurations bench.py times, checked plane by plane against the oracle.

  config 3 (headline): KukaButtonGymEnv-v0 ground_truth, 4096 envs, Philox env streams + device-sampled random-agent
      actions, io_device = 1 (torch-owned HBM buffers), chained srlhip_rollout launches over 2048 steps: every env
      crosses >= 2 auto-resets.  Bars (north star): |obs - oracle| <= 1e-4, actions / reward / done bit-exact.
  config 4: the same env with raw_pixels 64x64 through PixelStateVecEnv at 4096 envs: frames bit-exact against
      oracle/raster_oracle.c on the stepper's own state, encoder states against the float32 PyTorch forward (CPU)."""
import numpy as np
import pytest
import torch

from oracle import kuka_clib, raster_clib
from srlhip import _lib

pytestmark = pytest.mark.gpu
TOL = 1e-4


def test_kuka_bench_configuration_matches_oracle():
    n, T, chunks = 4096, 2048, (1024, 1024)               # bench.py: one 2048-step rollout per bench step; chained here
    cfg = _lib.default_config(_lib.ENV_KUKA_BUTTON)
    cfg.num_envs, cfg.seed0, cfg.rng_mode, cfg.auto_reset, cfg.io_device = n, 0, _lib.RNG_PHILOX, 1, 1
    h = _lib.Handle(cfg)
    dev = torch.device("cuda", 0)
    obs0 = torch.zeros((n, 3), dtype=torch.float32, device=dev)
    obs = torch.zeros((T, n, 3), dtype=torch.float32, device=dev)
    rew = torch.zeros((T, n), dtype=torch.float32, device=dev)
    done = torch.zeros((T, n), dtype=torch.uint8, device=dev)
    act = torch.zeros((T, n), dtype=torch.int32, device=dev)
    h.reset(obs_out=obs0.data_ptr())
    t = 0
    for c in chunks:
        h.rollout(c, out=(obs[t].data_ptr(), rew[t].data_ptr(), done[t].data_ptr(), act[t].data_ptr()))
        t += c
    ep_ret = torch.zeros((n,), dtype=torch.float32, device=dev)
    ep_len = torch.zeros((n,), dtype=torch.int32, device=dev)
    ep_fin = torch.zeros((n,), dtype=torch.int32, device=dev)
    h.episode_stats_device(ep_ret.data_ptr(), ep_len.data_ptr(), ep_fin.data_ptr())     # what the N>1 bench all-gathers
    h.sync()
    ora = kuka_clib.rollout(np.arange(n), T, actions=None, rng_mode=kuka_clib.RNG_PHILOX, trace=False)
    assert np.array_equal(act.cpu().numpy(), ora["actions"])
    assert np.array_equal(done.cpu().numpy(), ora["done"])
    assert np.array_equal(rew.cpu().numpy(), ora["reward"])
    assert np.abs(obs0.cpu().numpy() - ora["obs0"]).max() <= TOL
    assert np.abs(obs.cpu().numpy() - ora["obs"]).max() <= TOL
    assert np.abs(h.get_state(_lib.F_KUKA_Q).T - ora["final_state"][:, :7]).max() <= TOL
    fin = ora["ep_stats"][:, 2].astype(np.int32)
    assert fin.min() >= 2                                                  # every env crossed at least two resets
    assert np.array_equal(ep_fin.cpu().numpy(), fin)
    assert np.array_equal(ep_len.cpu().numpy(), ora["ep_stats"][:, 1].astype(np.int32))
    assert np.array_equal(ep_ret.cpu().numpy(), ora["ep_stats"][:, 0].astype(np.float32))
    ret64, length, fin2 = h.episode_stats()
    assert np.array_equal(ret64, ora["ep_stats"][:, 0]) and np.array_equal(fin2, fin)
    h.close()


def test_kuka_pixels_bench_configuration():
    from srlhip.pixel_env import PixelStateVecEnv
    from state_representation.models import SRLNeuralNetwork
    n = 4096
    torch.manual_seed(0)
    enc = SRLNeuralNetwork(3, cuda=True, img_shape=(64, 64), device=torch.device("cuda", 0))     # what bench_pixels builds
    assert enc.hip is not None
    env = PixelStateVecEnv("KukaButtonGymEnv-v0", n, enc, seed=0, img_shape=(64, 64))
    cpu = SRLNeuralNetwork(3, cuda=False, img_shape=(64, 64), state_dict=enc.model.state_dict(), backend="torch")

    def check(states):
        torch.cuda.synchronize()
        h = env.h
        st = np.concatenate([h.get_state(_lib.F_KUKA_Q).T, h.get_state(_lib.F_KUKA_BUTTON_Q)[0][:, None],
                             h.get_state(_lib.F_KUKA_BUTTON_XY).T], axis=1)
        frames = env.images.cpu().numpy()
        assert np.array_equal(frames, raster_clib.render(4, st, 64, 64, gripper_q=h.get_state(_lib.F_KUKA_GRIPPER_Q).T))          # bit-exact frames, all 4096 envs
        ref = cpu.getStates(frames).numpy()
        err = np.abs(states.cpu().numpy() - ref).max() / max(1.0, np.abs(ref).max())
        assert err <= 2e-5, err
        assert not enc.hip.overflow()

    check(env.reset())
    for k in range(40):
        states, rew, done = env.step()
        if k in (0, 39):
            check(states)
    # the stepper under the pixel pipeline is the same Philox random-agent rollout the ground-truth bench runs
    ora = kuka_clib.rollout(np.arange(n), 40, actions=None, rng_mode=kuka_clib.RNG_PHILOX, trace=False)
    assert np.abs(env.h.get_state(_lib.F_KUKA_Q).T - ora["final_state"][:, :7]).max() <= TOL
    assert np.array_equal(rew.cpu().numpy(), ora["reward"][-1]) and np.array_equal(done.cpu().numpy(), ora["done"][-1])
    env.close()
