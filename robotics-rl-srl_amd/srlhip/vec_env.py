"""HipVecEnv — the batched replacement for `[makeEnv(...)] -> DummyVecEnv | SubprocVecEnv`
(rl_baselines/utils.py:213-220): one handle on one GPU steps every env per call.

Duck-types stable_baselines.common.vec_env.VecEnv (2.5.0): num_envs,
observation_space, action_space, reset(), step_async(), step_wait(), step(),
close(), get_images(), render(); auto-reset on done; `None` actions allowed
(rl_baselines/evolution_strategies/ars.py:170); per-env Monitor CSV files and
info['episode'] like stable_baselines.bench.Monitor (environments/utils.py:54)."""
import json
import os
import time

import numpy as np

from . import _lib
from .envs import ENV_CLASSES, OBS_MODES
from .gym_compat import Box, Discrete

RNG_MODES = {"mt19937": _lib.RNG_MT19937, "philox": _lib.RNG_PHILOX}


def _env_kind(env_id):
    if env_id not in ENV_CLASSES:
        raise KeyError("{} is not stepped by libsrlhip (supported: {})".format(env_id, sorted(ENV_CLASSES)))
    return ENV_CLASSES[env_id].ENV_KIND


class HipVecEnv(object):
    def __init__(self, env_id, num_envs, seed=0, env_kwargs=None, device_id=0, first_env_id=0, rng_mode="mt19937",
                 log_dir=None, allow_early_resets=False):
        kw = dict(env_kwargs or {})
        self.env_id, self.num_envs, self.log_dir = env_id, int(num_envs), log_dir
        kind = _env_kind(env_id)
        cfg = _lib.default_config(kind)
        cfg.num_envs, cfg.device_id, cfg.first_env_id, cfg.seed0 = self.num_envs, device_id, first_env_id, int(seed)
        cfg.is_discrete = int(kw.get("is_discrete", True))
        cfg.random_target = int(kw.get("random_target", False))
        cfg.shape_reward = int(kw.get("shape_reward", False))
        cfg.force_down = int(kw.get("force_down", bool(cfg.force_down)))      # ctor default differs per env (Kuka2Button: False)
        cfg.action_repeat = int(kw.get("action_repeat", 1))
        cfg.action_joints = int(kw.get("action_joints", False))
        cfg.multi_view = int(bool(kw.get("multi_view", False)) or bool(kw.get("fpv", False)))   # Kuka: multi_view, Mobile: fpv
        if "max_distance" in kw:
            cfg.max_distance = float(kw["max_distance"])
        self.srl_model = kw.get("srl_model", "raw_pixels")
        if self.srl_model not in OBS_MODES:
            raise NotImplementedError("srl_model={!r}: learned SRL encoders plug in on top of raw_pixels".format(self.srl_model))
        cfg.obs_mode = OBS_MODES[self.srl_model]
        if "img_shape" in kw:                       # (H, W); the reference renders 224 x 224
            cfg.img_h, cfg.img_w = kw["img_shape"]
        cfg.rng_mode, cfg.auto_reset, cfg.io_device = RNG_MODES[rng_mode], 1, 0
        self.cfg = cfg
        self._h = _lib.Handle(cfg)
        if cfg.is_discrete:
            self.action_space = Discrete(self._h.num_actions)
        else:
            self.action_space = Box(low=-1, high=1, shape=(self._h.action_dim,), dtype=np.float32)
        if cfg.obs_mode == _lib.OBS_RAW_PIXELS:
            self.observation_space = Box(low=0, high=255, shape=(cfg.img_h, cfg.img_w, 6 if cfg.multi_view else 3), dtype=np.uint8)
        else:
            self.observation_space = Box(low=-np.inf, high=np.inf, shape=(self._h.obs_dim,), dtype=np.float32)
        self._actions = None
        self._infos = [{} for _ in range(self.num_envs)]     # reused between steps: one distinct dict per env
        self._dirty_infos = []
        self._n_finished = np.zeros(self.num_envs, np.int32)
        self._t_start = time.time()
        self._monitors = None
        if log_dir is not None:
            os.makedirs(log_dir, exist_ok=True)
            self._monitors = []
            for i in range(self.num_envs):
                f = open(os.path.join(log_dir, "{}.monitor.csv".format(first_env_id + i)), "wt")
                f.write("#%s\n" % json.dumps({"t_start": self._t_start, "env_id": env_id}))
                f.write("r,l,t\n")
                f.flush()
                self._monitors.append(f)

    # -- VecEnv API ----------------------------------------------------------------
    def reset(self):
        return self._h.reset()

    def step_async(self, actions):
        if self.cfg.is_discrete:
            if isinstance(actions, np.ndarray) and actions.dtype != object:      # fast path: no `None` entries possible
                self._actions = np.ascontiguousarray(actions, dtype=np.int32).reshape(self.num_envs)
            else:
                self._actions = np.array([-1 if a is None else int(a) for a in actions], dtype=np.int32)
            if self._actions.size and (self._actions.min() < -1 or self._actions.max() >= self._h.num_actions):
                # the reference indexes a per-action list (`[-dv, dv, 0, 0, 0, 0][action]`): IndexError in the worker
                raise IndexError("discrete action out of range [0, {}) (None/-1 = no-op)".format(self._h.num_actions))
        else:
            if any(a is None for a in actions):
                raise NotImplementedError("None actions need a discrete action space")
            self._actions = np.asarray(actions, dtype=np.float32).reshape(self.num_envs, self._h.action_dim)

    def step_wait(self):
        obs, rew, done = self._h.step(self._actions)
        dones = done.astype(bool)
        for i in self._dirty_infos:                      # entries that carried an 'episode' record last step
            self._infos[i] = {}
        self._dirty_infos = []
        infos = self._infos
        if dones.any():
            ret, length, fin = self._h.episode_stats()
            t = round(time.time() - self._t_start, 6)
            for i in np.nonzero(dones)[0]:
                ep = {"r": round(float(ret[i]), 6), "l": int(length[i]), "t": t}
                infos[i] = {"episode": ep}
                self._dirty_infos.append(int(i))
                if self._monitors is not None:
                    self._monitors[i].write("{},{},{}\n".format(ep["r"], ep["l"], ep["t"]))
                    self._monitors[i].flush()
            self._n_finished = fin
        return obs, rew, dones, list(infos)

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def rollout(self, n_steps, actions=None):
        """Fused device-side rollout (no per-step host round trip): dict of [T][N] planes."""
        return self._h.rollout(n_steps, actions=actions)

    def get_images(self):
        return list(self._h.render())

    def render(self, mode="human"):
        return np.array([])

    def seed(self, seed):
        self._h.seed(int(seed) + self.cfg.first_env_id + np.arange(self.num_envs, dtype=np.int64))

    def episode_returns(self):
        """Per-env return/length of the last finished episode and the number of finished episodes."""
        return self._h.episode_stats()

    def close(self):
        if self._monitors is not None:
            for f in self._monitors:
                f.close()
            self._monitors = None
        if self._h is not None:
            self._h.close()
            self._h = None

    @property
    def unwrapped(self):
        return self
