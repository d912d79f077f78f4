"""Device-resident VecEnv stack and the batched ARS evaluation.  The tensor wrappers are device
agnostic, so their arithmetic is checked on CPU tensors against the numpy shims; the GPU tests run the real handle."""
import os
import sys
import types

import numpy as np
import pytest
import torch

from srlhip.device_env import DeviceVecFrameStack, DeviceVecNormalize
from srlhip.gym_compat import Box, Discrete
from srlhip import vec_wrappers


class FakeTensorEnv(object):
    """deterministic stand-in: obs/reward/done scripted from a seeded numpy stream (CPU tensors)"""

    def __init__(self, n=6, d=3, seed=0):
        self.num_envs, self.device = n, torch.device("cpu")
        self.observation_space = Box(low=-np.inf, high=np.inf, shape=(d,), dtype=np.float32)
        self.action_space = Discrete(4)
        self.rng, self.d = np.random.RandomState(seed), d

    def _next(self):
        return (self.rng.normal(size=(self.num_envs, self.d)).astype(np.float32) * 3 + 1,
                self.rng.normal(size=self.num_envs).astype(np.float32), self.rng.rand(self.num_envs) < 0.15)

    def reset(self):
        return torch.from_numpy(self._next()[0])

    def step(self, actions):
        o, r, d = self._next()
        return torch.from_numpy(o), torch.from_numpy(r), torch.from_numpy(d.astype(np.uint8))

    # numpy VecEnv duck type for the reference-style shims
    def step_async(self, a):
        pass

    def step_wait(self):
        o, r, d = self._next()
        return o, r, d, [{} for _ in range(self.num_envs)]


class FakeNumpyEnv(FakeTensorEnv):
    def reset(self):
        return self._next()[0]


def test_frame_stack_and_normalize_match_the_numpy_shims():
    a = DeviceVecNormalize(DeviceVecFrameStack(FakeTensorEnv(seed=3), 4), norm_obs=True, norm_reward=True)
    b = vec_wrappers.VecNormalize(vec_wrappers.VecFrameStack(FakeNumpyEnv(seed=3), 4), norm_obs=True, norm_reward=True)
    oa, ob = a.reset(), b.reset()
    assert oa.shape == (6, 12) and np.allclose(oa.numpy(), ob, atol=1e-5)
    for t in range(60):
        oa, ra, da = a.step(torch.zeros(6, dtype=torch.int32))
        ob, rb, db, _ = b.step(None)
        assert np.allclose(oa.numpy(), ob, atol=2e-5) and np.allclose(ra.numpy(), rb, atol=2e-5) and np.array_equal(da.numpy() != 0, db)
    assert np.allclose(a.obs_rms.mean.numpy(), b.obs_rms.mean) and np.allclose(a.obs_rms.var.numpy(), b.obs_rms.var)
    assert np.allclose(a.get_original_obs().numpy(), b.get_original_obs())


def test_batched_policy_evaluation_equals_the_reference_loop():
    """ars.py:160-172 evaluates env 2k with M + noise*delta_k and env 2k+1 with M - noise*delta_k, None when done."""
    from rl_baselines.evolution_strategies.ars import ARSModel
    rng = np.random.RandomState(0)
    P, D, A, noise = 5, 4, 6, 0.02
    M, delta = rng.normal(size=(D, A)), rng.normal(size=(P, D, A))
    obs = rng.normal(size=(2 * P, D)).astype(np.float32)
    done = rng.rand(2 * P) < 0.3
    model = ARSModel()
    model.M, model.continuous_actions, model.deterministic = M, False, True
    expect = []
    for k in range(P):
        for direction in range(2):
            if not done[k * 2 + direction]:
                d = noise * delta[k] if direction == 0 else -noise * delta[k]
                expect.append(int(model.getAction([obs[k * 2 + direction].reshape(-1)], delta=d)[0]))
            else:
                expect.append(-1)
    got = ARSModel.batched_actions(torch.from_numpy(obs), torch.from_numpy(M), torch.from_numpy(delta), noise,
                                   torch.from_numpy(~done), False, True)
    assert got.dtype == torch.int32 and got.tolist() == expect
    cont = ARSModel.batched_actions(torch.from_numpy(obs), torch.from_numpy(M), torch.from_numpy(delta), noise,
                                    torch.from_numpy(~done), True, True)
    model.continuous_actions = True
    ref0 = model.getAction([obs[0]], delta=noise * delta[0])[0]
    assert cont.shape == (2 * P, A) and (done[0] or np.allclose(cont[0].numpy(), ref0, atol=1e-5))
    p = model.getActionProba(obs[:2], delta=0)
    assert p.shape == (2, A)


@pytest.mark.gpu
def test_device_env_steps_like_the_host_vec_env():
    from srlhip.device_env import DeviceVecEnv
    from srlhip.vec_env import HipVecEnv
    n, T = 64, 300
    kw = {"srl_model": "ground_truth"}
    dev = DeviceVecEnv("MobileRobotGymEnv-v0", n, seed=5, env_kwargs=kw)
    host = HipVecEnv("MobileRobotGymEnv-v0", n, seed=5, env_kwargs=kw, rng_mode="philox")     # DeviceVecEnv's default stream family
    od, oh = dev.reset(), host.reset()
    assert od.is_cuda and np.array_equal(od.cpu().numpy(), oh)
    acts = np.random.RandomState(1).randint(-1, 4, size=(T, n)).astype(np.int32)
    with torch.cuda.stream(dev.torch_stream):                         # ordered with the stepper: no host syncs inside step()
        for t in range(T):
            o, r, d = dev.step(torch.from_numpy(acts[t]).to(dev.device, non_blocking=False))
            if t % 97 == 0 or t == T - 1:
                oh, rh, dh, _ = host.step([None if a < 0 else int(a) for a in acts[t]])
                assert np.array_equal(o.cpu().numpy(), oh) and np.array_equal(r.cpu().numpy(), rh) and np.array_equal(d.cpu().numpy() != 0, dh)
            else:
                host.step([None if a < 0 else int(a) for a in acts[t]])
    o2, r2, d2 = dev.step(torch.from_numpy(acts[0]).to(dev.device))     # default stream: the synchronising fallback path
    assert o2.shape == (n, 2)
    dev.close(); host.close()


@pytest.mark.gpu
def test_ars_trains_on_the_device():
    from rl_baselines.evolution_strategies.ars import ARSModel
    args = types.SimpleNamespace(env="MobileRobot1DGymEnv-v0", seed=0, num_population=16, top_population=4, step_size=0.05,
                                 exploration_noise=0.05, max_step_amplitude=10, deterministic=True, algo_type="v2",
                                 num_stack=1, srl_model="ground_truth", continuous_actions=False,
                                 num_timesteps=16 * 16 * 270)       # reference arithmetic (ars.py:150): 2 * T / P counter units, +P per env step
    seen = []
    model = ARSModel().train(args, callback=lambda l, g: seen.append(l["step"]),
                             env_kwargs={"srl_model": "ground_truth", "shape_reward": True})
    assert model.M.shape == (1, 2) and np.isfinite(model.M).all() and np.abs(model.M).max() > 0
    assert len(model.history) >= 2 and len(seen) >= 2 * 251 and seen[0] == 16
    import tempfile, os
    path = os.path.join(tempfile.mkdtemp(), "ars.pkl")
    model.save(path)
    again = ARSModel.load(path)
    assert np.array_equal(again.M, model.M) and again.getAction(np.ones((3, 1))).shape == (3,)


@pytest.mark.gpu
def test_device_normalize_statistics_round_trip(tmp_path):
    """ARS v2 trains on normalised observations: the running statistics have to survive save / load, in the file layout of
    stable_baselines' VecNormalize (obs_rms.pkl / ret_rms.pkl), interchangeable with the host-side wrapper."""
    from srlhip.device_env import DeviceVecEnv, DeviceVecNormalize
    from srlhip.vec_env import HipVecEnv
    from srlhip.vec_wrappers import VecNormalize
    kw = {"srl_model": "ground_truth"}
    env = DeviceVecNormalize(DeviceVecEnv("MobileRobotGymEnv-v0", 32, seed=1, env_kwargs=kw), norm_reward=False)
    with torch.cuda.stream(env.torch_stream):
        env.reset()
        for _ in range(40):
            env.step(torch.randint(0, 4, (32,), dtype=torch.int32, device=env.device))
    env.save_running_average(str(tmp_path))
    twin = DeviceVecNormalize(DeviceVecEnv("MobileRobotGymEnv-v0", 32, seed=1, env_kwargs=kw), norm_reward=False, training=False)
    twin.load_running_average(str(tmp_path))
    assert torch.equal(twin.obs_rms.mean, env.obs_rms.mean) and torch.equal(twin.obs_rms.var, env.obs_rms.var)
    assert twin.obs_rms.count == env.obs_rms.count
    host = VecNormalize(HipVecEnv("MobileRobotGymEnv-v0", 32, seed=1, env_kwargs=kw, rng_mode="philox"), norm_reward=False, training=False)
    host.load_running_average(str(tmp_path))
    assert np.array_equal(host.obs_rms.mean, env.obs_rms.mean.cpu().numpy())
    o_dev, o_host = twin.reset(), host.reset()
    assert np.allclose(o_dev.cpu().numpy(), o_host, atol=1e-6)
    env.close(); twin.close(); host.close()


@pytest.mark.gpu
def test_vec_env_monitor_files_and_early_reset_policy(tmp_path):
    """4096 envs with a log_dir: one monitor file per env without holding 4096 descriptors open; a second reset() in the
    middle of an episode is refused unless allow_early_resets (stable_baselines.bench.Monitor semantics)."""
    from srlhip.vec_env import HipVecEnv
    env = HipVecEnv("MobileRobotGymEnv-v0", 4096, seed=0, env_kwargs={"srl_model": "ground_truth"}, log_dir=str(tmp_path))
    env.reset()
    acts = np.zeros(4096, np.int32)
    for _ in range(251):
        obs, rew, done, infos = env.step(acts)
    assert done.all() and infos[4095]["episode"]["l"] == 251
    assert open(str(tmp_path / "4095.monitor.csv")).read().splitlines()[2].split(",")[1] == "251"
    env.reset()                                   # right after the episode end: no running episode
    env.step(acts)
    with pytest.raises(RuntimeError):
        env.reset()
    env.close()
    env = HipVecEnv("MobileRobotGymEnv-v0", 8, seed=0, env_kwargs={"srl_model": "ground_truth"}, allow_early_resets=True)
    env.reset(); env.step(np.zeros(8, np.int32)); env.reset()
    env.close()


def test_cma_es_strategy_minimises_a_quadratic():
    """the restated (mu/mu_w, lambda)-CMA-ES on a rotated ill-conditioned quadratic (CPU tensors)"""
    from rl_baselines.evolution_strategies.cma_es import CMAES, BatchedMLP
    n = 12
    rs = np.random.RandomState(0)
    Q = np.linalg.qr(rs.randn(n, n))[0]
    A = torch.as_tensor(Q @ np.diag(np.logspace(0, 3, n)) @ Q.T)
    es = CMAES(n * [1.0], 0.5, 16, torch.device("cpu"), seed=1)
    for _ in range(400):
        x = es.ask()
        es.tell(x, ((x @ A) * x).sum(1))
    assert es.fbest < 1e-12 and float((es.mean ** 2).sum()) < 1e-12           # log-linear convergence on a 1e3-conditioned bowl
    mlp = BatchedMLP(3, 2, hidden=5)                                     # parameter layout = nn.Module.parameters() order
    net = torch.nn.Sequential(torch.nn.Linear(3, 5), torch.nn.ReLU(), torch.nn.Linear(5, 2)).double()
    flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    obs = torch.randn(4, 3, dtype=torch.float64)
    assert torch.allclose(mlp.forward(flat.unsqueeze(0).expand(4, -1), obs), net(obs))


@pytest.mark.gpu
def test_cma_es_trains_on_the_device(tmp_path):
    from rl_baselines.evolution_strategies.cma_es import CMAESModel
    args = types.SimpleNamespace(env="MobileRobot1DGymEnv-v0", seed=0, num_population=16, mu=0.0, sigma=0.14, cuda=True,
                                 deterministic=True, num_stack=1, srl_model="ground_truth", continuous_actions=False,
                                 num_timesteps=16 * 251 * 4, log_dir=str(tmp_path))
    seen = []
    model = CMAESModel().train(args, callback=lambda l, g: seen.append(l["k"]),
                               env_kwargs={"srl_model": "ground_truth", "shape_reward": True})
    assert len(model.history) >= 4 and len(seen) >= 4 * 251 and np.isfinite(model.best_model).all()
    assert model.best_model.shape == (model.policy.n_params,) and np.abs(model.best_model).max() > 0
    assert os.path.exists(os.path.join(str(tmp_path), "obs_rms.pkl"))
    path = os.path.join(str(tmp_path), "cma.pkl")
    model.save(path)
    again = CMAESModel.load(path)
    assert np.array_equal(again.best_model, model.best_model) and again.getAction(np.ones((3, 1))).shape == (3,)
    assert np.allclose(again.getActionProba(np.ones((2, 1))).sum(1), 1.0)
