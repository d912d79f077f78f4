"""Device-resident VecEnv stack: observations, rewards, dones and actions are torch tensors in HBM
(io_device = 1), so a policy that also lives on the GPU never round-trips through the host.

    DeviceVecEnv         the libsrlhip handle behind a tensor-in / tensor-out step()   (rl_baselines/utils.py:213-220)
    DeviceVecFrameStack  stable-baselines VecFrameStack on tensors                      (rl_baselines/utils.py:222)
    DeviceVecNormalize   stable-baselines VecNormalize (running mean / var) on tensors  (rl_baselines/utils.py:223-227)

Stream discipline: the handle owns a HIP stream (srlhip_stream).  `env.torch_stream` wraps it as a
torch.cuda.ExternalStream; code that runs under `with torch.cuda.stream(env.torch_stream):` is ordered with the
stepper's kernels and needs no host synchronisation at all.  When the caller is on another stream, step()/reset()
fall back to two host-side stream synchronisations per call."""
import numpy as np
import torch

from . import _lib
from .envs import ENV_CLASSES, OBS_MODES
from .gym_compat import Box, Discrete

from .vec_env import RNG_MODES, default_rng_mode


class DeviceVecEnv(object):
    def __init__(self, env_id, num_envs, seed=0, env_kwargs=None, device_id=0, first_env_id=0, rng_mode=None):
        kw = dict(env_kwargs or {})
        rng_mode = rng_mode or default_rng_mode("device")
        cfg = _lib.default_config(ENV_CLASSES[env_id].ENV_KIND)
        cfg.num_envs, cfg.device_id, cfg.first_env_id, cfg.seed0 = int(num_envs), device_id, first_env_id, int(seed)
        for name in ("is_discrete", "random_target", "shape_reward", "force_down", "action_repeat", "action_joints"):
            if name in kw:
                setattr(cfg, name, int(kw[name]))
        if "max_distance" in kw:
            cfg.max_distance = float(kw["max_distance"])
        srl_model = kw.get("srl_model", "ground_truth")
        if srl_model not in OBS_MODES or srl_model == "raw_pixels":
            raise NotImplementedError("DeviceVecEnv serves the state observations; use PixelStateVecEnv for raw_pixels")
        cfg.obs_mode = OBS_MODES[srl_model]
        cfg.rng_mode, cfg.auto_reset, cfg.io_device = RNG_MODES[rng_mode], 1, 1
        self.cfg, self.env_id, self.num_envs = cfg, env_id, int(num_envs)
        self.h = _lib.Handle(cfg)
        self.device = torch.device("cuda", device_id)
        self.torch_stream = torch.cuda.ExternalStream(self.h.stream(), device=self.device)
        n = self.num_envs
        self.obs = torch.zeros((n, self.h.obs_dim), dtype=torch.float32, device=self.device)
        self.rewards = torch.zeros((n,), dtype=torch.float32, device=self.device)
        self.dones = torch.zeros((n,), dtype=torch.uint8, device=self.device)
        if cfg.is_discrete:
            self.action_space = Discrete(self.h.num_actions)
        else:
            self.action_space = Box(low=-1, high=1, shape=(self.h.action_dim,), dtype=np.float32)
        self.observation_space = Box(low=-np.inf, high=np.inf, shape=(self.h.obs_dim,), dtype=np.float32)

    def _on_env_stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream == self.torch_stream.cuda_stream

    def reset(self):
        ordered = self._on_env_stream()
        if not ordered:
            torch.cuda.current_stream(self.device).synchronize()
        self.h.reset(obs_out=self.obs.data_ptr())
        if not ordered:
            self.h.sync()
        return self.obs

    def step(self, actions):
        """actions: int32 tensor [N] (-1 == the reference's None) or float32 tensor [N, action_dim], on this device.
        Returns (obs, rewards, dones): tensors owned by the env, overwritten by the next call."""
        want = torch.int32 if self.cfg.is_discrete else torch.float32
        assert actions.is_cuda and actions.dtype == want and actions.is_contiguous(), (actions.dtype, actions.device)
        ordered = self._on_env_stream()
        if not ordered:
            torch.cuda.current_stream(self.device).synchronize()
        self.h.step(actions.data_ptr(), out=(self.obs.data_ptr(), self.rewards.data_ptr(), self.dones.data_ptr()))
        if not ordered:
            self.h.sync()
        return self.obs, self.rewards, self.dones

    def episode_stats(self):
        return self.h.episode_stats()

    def close(self):
        self.h.close()


class DeviceVecEnvWrapper(object):
    def __init__(self, venv, observation_space=None):
        self.venv, self.num_envs = venv, venv.num_envs
        self.observation_space = observation_space or venv.observation_space
        self.action_space = venv.action_space

    def __getattr__(self, name):
        return getattr(self.venv, name)

    def close(self):
        return self.venv.close()


class DeviceVecFrameStack(DeviceVecEnvWrapper):
    """stable_baselines.common.vec_env.VecFrameStack on tensors: newest frame last, a finished env's stack is cleared."""

    def __init__(self, venv, n_stack):
        wos = venv.observation_space
        low, high = np.repeat(wos.low, n_stack, axis=-1), np.repeat(wos.high, n_stack, axis=-1)
        super(DeviceVecFrameStack, self).__init__(venv, Box(low=low, high=high, dtype=wos.dtype))
        self.n_stack, self.stacked = n_stack, None

    def _push(self, obs, dones=None):
        last = obs.shape[-1]
        if self.stacked is None:
            self.stacked = torch.zeros((obs.shape[0], last * self.n_stack), dtype=obs.dtype, device=obs.device)
        self.stacked = torch.roll(self.stacked, shifts=-last, dims=-1)
        if dones is not None:
            self.stacked = self.stacked * (dones == 0).to(obs.dtype).unsqueeze(-1)
        self.stacked[..., -last:] = obs
        return self.stacked

    def reset(self):
        obs = self.venv.reset()
        self.stacked = None
        return self._push(obs)

    def step(self, actions):
        obs, rew, done = self.venv.step(actions)
        return self._push(obs, done), rew, done


class RunningMeanStd(object):
    """stable_baselines.common.running_mean_std.RunningMeanStd (parallel-variance batch update) on tensors"""

    def __init__(self, shape, device, epsilon=1e-4):
        self.mean = torch.zeros(shape, dtype=torch.float64, device=device)
        self.var = torch.ones(shape, dtype=torch.float64, device=device)
        self.count = epsilon

    def update(self, x):
        x = x.to(torch.float64)
        bmean, bvar, bcount = x.mean(0), x.var(0, unbiased=False), x.shape[0]
        delta, tot = bmean - self.mean, self.count + bcount
        m2 = self.var * self.count + bvar * bcount + delta * delta * (self.count * bcount / tot)
        self.mean, self.var, self.count = self.mean + delta * (bcount / tot), m2 / tot, tot


class DeviceVecNormalize(DeviceVecEnvWrapper):
    """stable_baselines VecNormalize: obs -> clip((obs - mean) / sqrt(var + eps)), optional return-based reward scaling."""

    def __init__(self, venv, training=True, norm_obs=True, norm_reward=True, clip_obs=10.0, clip_reward=10.0, gamma=0.99,
                 epsilon=1e-8):
        super(DeviceVecNormalize, self).__init__(venv)
        dev = venv.device
        self.obs_rms = RunningMeanStd(self.observation_space.shape, dev)
        self.ret_rms = RunningMeanStd((), dev)
        self.ret = torch.zeros(self.num_envs, dtype=torch.float64, device=dev)
        self.training, self.norm_obs, self.norm_reward = training, norm_obs, norm_reward
        self.clip_obs, self.clip_reward, self.gamma, self.epsilon = clip_obs, clip_reward, gamma, epsilon
        self.old_obs = None

    # same files as stable_baselines' VecNormalize (and srlhip.vec_wrappers.VecNormalize): {path}/obs_rms.pkl, ret_rms.pkl with
    # host-side (mean, var, count) objects, so statistics move freely between the host and the device wrapper
    def save_running_average(self, path):
        import pickle
        from .vec_wrappers import _RunningMeanStd as HostRms
        for rms, name in ((self.obs_rms, "obs_rms"), (self.ret_rms, "ret_rms")):
            host = HostRms(tuple(rms.mean.shape))
            host.mean, host.var, host.count = rms.mean.cpu().numpy(), rms.var.cpu().numpy(), float(rms.count)
            with open("{}/{}.pkl".format(path, name), "wb") as f:
                pickle.dump(host, f)

    def load_running_average(self, path):
        import pickle
        for rms, name in ((self.obs_rms, "obs_rms"), (self.ret_rms, "ret_rms")):
            with open("{}/{}.pkl".format(path, name), "rb") as f:
                host = pickle.load(f)
            rms.mean = torch.as_tensor(np.asarray(host.mean), dtype=torch.float64, device=rms.mean.device).reshape(rms.mean.shape)
            rms.var = torch.as_tensor(np.asarray(host.var), dtype=torch.float64, device=rms.var.device).reshape(rms.var.shape)
            rms.count = float(host.count)

    saveRunningAverage, loadRunningAverage = save_running_average, load_running_average

    def _obfilt(self, obs):
        if not self.norm_obs:
            return obs
        if self.training:
            self.obs_rms.update(obs)
        out = (obs.to(torch.float64) - self.obs_rms.mean) / torch.sqrt(self.obs_rms.var + self.epsilon)
        return torch.clamp(out, -self.clip_obs, self.clip_obs).to(torch.float32)

    def get_original_obs(self):
        return self.old_obs

    def reset(self):
        obs = self.venv.reset()
        self.old_obs = obs
        self.ret.zero_()
        return self._obfilt(obs)

    def step(self, actions):
        obs, rew, done = self.venv.step(actions)
        self.old_obs = obs
        self.ret = self.ret * self.gamma + rew.to(torch.float64)
        out = self._obfilt(obs)
        if self.norm_reward:
            if self.training:
                self.ret_rms.update(self.ret)
            rew = torch.clamp(rew.to(torch.float64) / torch.sqrt(self.ret_rms.var + self.epsilon), -self.clip_reward,
                              self.clip_reward).to(torch.float32)
        self.ret = self.ret * (done == 0).to(torch.float64)
        return out, rew, done
