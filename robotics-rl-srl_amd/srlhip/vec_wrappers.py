"""Numpy versions of the two stable-baselines wrappers createEnvs stacks on the
vec env (rl_baselines/utils.py:222-227).  Used only when stable_baselines is not
importable; both need nothing but the VecEnv duck type."""
import numpy as np

from .gym_compat import Box


class VecEnvWrapper(object):
    def __init__(self, venv, observation_space=None):
        self.venv = venv
        self.num_envs = venv.num_envs
        self.observation_space = observation_space or venv.observation_space
        self.action_space = venv.action_space

    def step_async(self, actions):
        self.venv.step_async(actions)

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def close(self):
        return self.venv.close()

    def render(self, *a, **k):
        return self.venv.render(*a, **k)

    def get_images(self):
        return self.venv.get_images()

    def __getattr__(self, name):
        return getattr(self.venv, name)


class VecFrameStack(VecEnvWrapper):
    def __init__(self, venv, n_stack):
        wos = venv.observation_space
        low, high = np.repeat(wos.low, n_stack, axis=-1), np.repeat(wos.high, n_stack, axis=-1)
        self.n_stack = n_stack
        self.stackedobs = np.zeros((venv.num_envs,) + low.shape, low.dtype)
        super(VecFrameStack, self).__init__(venv, Box(low=low, high=high, dtype=wos.dtype))

    def step_wait(self):
        obs, rews, news, infos = self.venv.step_wait()
        last = obs.shape[-1]
        self.stackedobs = np.roll(self.stackedobs, shift=-last, axis=-1)
        for i, new in enumerate(news):
            if new:
                self.stackedobs[i] = 0
        self.stackedobs[..., -last:] = obs
        return self.stackedobs, rews, news, infos

    def reset(self):
        obs = self.venv.reset()
        self.stackedobs[...] = 0
        self.stackedobs[..., -obs.shape[-1]:] = obs
        return self.stackedobs


class _RunningMeanStd(object):
    def __init__(self, shape):
        self.mean, self.var, self.count = np.zeros(shape, np.float64), np.ones(shape, np.float64), 1e-4

    def update(self, x):
        bm, bv, bc = x.mean(axis=0), x.var(axis=0), x.shape[0]
        delta, tot = bm - self.mean, self.count + bc
        self.mean = self.mean + delta * bc / tot
        self.var = (self.var * self.count + bv * bc + delta ** 2 * self.count * bc / tot) / tot
        self.count = tot


class VecNormalize(VecEnvWrapper):
    def __init__(self, venv, training=True, norm_obs=True, norm_reward=True, clip_obs=10., clip_reward=10., gamma=0.99,
                 epsilon=1e-8):
        super(VecNormalize, self).__init__(venv)
        self.obs_rms, self.ret_rms = _RunningMeanStd(self.observation_space.shape), _RunningMeanStd(())
        self.clip_obs, self.clip_reward, self.gamma, self.epsilon = clip_obs, clip_reward, gamma, epsilon
        self.training, self.norm_obs, self.norm_reward = training, norm_obs, norm_reward
        self.ret = np.zeros(self.num_envs)
        self.old_obs = np.array([])

    def _normalize_observation(self, obs):
        if not self.norm_obs:
            return obs
        if self.training:
            self.obs_rms.update(obs)
        return np.clip((obs - self.obs_rms.mean) / np.sqrt(self.obs_rms.var + self.epsilon), -self.clip_obs, self.clip_obs)

    def step_wait(self):
        obs, rews, news, infos = self.venv.step_wait()
        self.ret = self.ret * self.gamma + rews
        self.old_obs = obs
        obs = self._normalize_observation(obs)
        if self.norm_reward:
            if self.training:
                self.ret_rms.update(self.ret)
            rews = np.clip(rews / np.sqrt(self.ret_rms.var + self.epsilon), -self.clip_reward, self.clip_reward)
        self.ret[news] = 0
        return obs, rews, news, infos

    def get_original_obs(self):
        return self.old_obs

    def reset(self):
        obs = self.venv.reset()
        self.old_obs = obs
        self.ret = np.zeros(self.num_envs)
        return self._normalize_observation(obs)

    def save_running_average(self, path):
        import pickle
        for rms, name in zip([self.obs_rms, self.ret_rms], ['obs_rms', 'ret_rms']):
            with open("{}/{}.pkl".format(path, name), 'wb') as f:
                pickle.dump(rms, f)

    def load_running_average(self, path):
        import pickle
        for name in ['obs_rms', 'ret_rms']:
            with open("{}/{}.pkl".format(path, name), 'rb') as f:
                setattr(self, name, pickle.load(f))
