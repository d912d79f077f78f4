"""HipVecEnv — the batched replacement for `[makeEnv(...)] -> DummyVecEnv | SubprocVecEnv`
(rl_baselines/utils.py:213-220): one handle on one GPU steps every env per call.

Duck-types stable_baselines.common.vec_env.VecEnv (2.5.0): num_envs,
observation_space, action_space, reset(), step_async(), step_wait(), step(),
close(), get_images(), render(); auto-reset on done; `None` actions allowed
(rl_baselines/evolution_strategies/ars.py:170); per-env Monitor CSV files and
info['episode'] like stable_baselines.bench.Monitor (environments/utils.py:54).

Learned SRL models (state_representation/registry.py: SRLType.SRL — autoencoder, robotic_priors, ...): the reference
runs ONE encoder process that every env queries through queues (MultiprocessSRLModel, rl_baselines/utils.py:162-191).
Here the stepper renders all frames into HBM, the encoder (state_representation.models: fused HIP kernel for 64x64x3
frames, PyTorch-ROCm otherwise) maps the batch to states on the same device, and only [N][state_dim] floats cross
PCIe: the observation the policy sees is `srl_model.getState(render())`, as in the reference."""
import json
import os
import time

import numpy as np

from . import _lib
from .envs import ENV_CLASSES, OBS_MODES
from .gym_compat import Box, Discrete

# "mt19937": the reference's own streams — every env draws from a device-resident numpy RandomState seeded like
# gym.utils.seeding.np_random(seed + rank) (srl_env.py:71-78, environments/utils.py:52), so noise / reset draws are the
# reference's bit for bit on identical seeds.  It is the default of the reference-surface entry points (HipVecEnv, i.e.
# rl_baselines.utils.createEnvs / rl_baselines.train / dataset_generator): a drop-in reproduces the reference's draws with
# no environment variable.  "philox": counter-based per-env streams, the fast path of both steppers (the MT19937
# generators' 624-word regenerations cost the Kuka stepper ~15 %); default of the device-resident stack (DeviceVecEnv,
# PixelStateVecEnv, bench.py), which has no reference stream to reproduce.  SRLHIP_RNG_MODE overrides both defaults.
RNG_MODES = {"mt19937": _lib.RNG_MT19937, "philox": _lib.RNG_PHILOX}


def default_rng_mode(surface="reference"):
    mode = os.environ.get("SRLHIP_RNG_MODE", "mt19937" if surface == "reference" else "philox")
    if mode not in RNG_MODES:
        raise ValueError("SRLHIP_RNG_MODE must be one of {}".format(sorted(RNG_MODES)))
    return mode


def _env_kind(env_id):
    if env_id not in ENV_CLASSES:
        raise KeyError("{} is not stepped by libsrlhip (supported: {})".format(env_id, sorted(ENV_CLASSES)))
    return ENV_CLASSES[env_id].ENV_KIND


class HipVecEnv(object):
    def __init__(self, env_id, num_envs, seed=0, env_kwargs=None, device_id=0, first_env_id=0, rng_mode=None,
                 log_dir=None, allow_early_resets=False, encoder=None):
        kw = dict(env_kwargs or {})
        rng_mode = rng_mode or default_rng_mode()
        self.allow_early_resets, self._was_reset = bool(allow_early_resets), False
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
        if "img_shape" in kw:                       # (H, W); the reference renders 224 x 224
            cfg.img_h, cfg.img_w = kw["img_shape"]
        self._enc = None
        if self.srl_model not in OBS_MODES:
            # a learned SRL model: pixels stay on the device, the encoder turns them into the observation
            from state_representation.registry import registered_srl, SRLType
            if registered_srl.get(self.srl_model, (None,))[0] is not SRLType.SRL:
                raise KeyError("unknown srl_model {!r}".format(self.srl_model))
            if encoder is None:
                from state_representation.models import loadSRLModel
                encoder = loadSRLModel(kw.get("srl_model_path"), cuda=True, state_dim=kw.get("state_dim"),
                                       img_shape=(cfg.img_h, cfg.img_w), n_channels=6 if cfg.multi_view else 3)
            self._enc = encoder
            cfg.obs_mode, cfg.io_device = _lib.OBS_RAW_PIXELS, 1
        else:
            cfg.obs_mode, cfg.io_device = OBS_MODES[self.srl_model], 0
        cfg.rng_mode, cfg.auto_reset = RNG_MODES[rng_mode], 1
        self.cfg = cfg
        self._h = _lib.Handle(cfg)
        if self._enc is not None:
            import torch
            dev = torch.device("cuda", device_id)
            ch = 6 if cfg.multi_view else 3
            self._t = {"images": torch.zeros((self.num_envs, cfg.img_h, cfg.img_w, ch), dtype=torch.uint8, device=dev),
                       "rew": torch.zeros((self.num_envs,), dtype=torch.float32, device=dev),
                       "done": torch.zeros((self.num_envs,), dtype=torch.uint8, device=dev),
                       "act": torch.zeros((self.num_envs,) if cfg.is_discrete else (self.num_envs, self._h.action_dim),
                                          dtype=torch.int32 if cfg.is_discrete else torch.float32, device=dev)}
            self.state_dim = int(self._enc.state_dim)
        if cfg.is_discrete:
            self.action_space = Discrete(self._h.num_actions)
        else:
            self.action_space = Box(low=-1, high=1, shape=(self._h.action_dim,), dtype=np.float32)
        if self._enc is not None:
            self.observation_space = Box(low=-np.inf, high=np.inf, shape=(self.state_dim,), dtype=np.float32)
        elif cfg.obs_mode == _lib.OBS_RAW_PIXELS:
            self.observation_space = Box(low=0, high=255, shape=(cfg.img_h, cfg.img_w, 6 if cfg.multi_view else 3), dtype=np.uint8)
        else:
            self.observation_space = Box(low=-np.inf, high=np.inf, shape=(self._h.obs_dim,), dtype=np.float32)
        self._actions = None
        # infos of a step in which no episode ended: ONE immutable tuple of per-env empty dicts, handed out as is (SubprocVecEnv
        # returns a tuple too); a step with episode records gets its own list
        self._quiet_infos = tuple({} for _ in range(self.num_envs))
        self._n_finished = np.zeros(self.num_envs, np.int32)
        # per-step fast path of the ground-truth observation modes (host-pointer handle): actions are copied into ONE bound array, the
        # foreign call is bound once (Handle.step_fn), Monitor's (r, l) are read from the handle's mapped record planes
        # (srlhip_episode_records) — no device copy, no second call, when an episode ends
        self._fast = None
        if self._enc is None:
            act = np.zeros((self.num_envs,) if cfg.is_discrete else (self.num_envs, self._h.action_dim), np.int32 if cfg.is_discrete else np.float32)
            call, obs, rew, done = self._h.step_fn(act)
            ret, length = self._h.episode_records()
            self._fast = {"act": act, "call": call, "obs": obs, "rew": rew, "done": done, "ret": ret, "len": length}
        self._mon_rows, self._mon_count, self._mon_flushed = {}, 0, time.time()
        self._t_start = time.time()
        # bench.Monitor files, one per env like the reference (environments/utils.py:54).  They are NOT kept open: at the
        # batch sizes this env is meant for (4096+) that would exceed RLIMIT_NOFILE; rows are appended when episodes end.
        self._monitors = None
        if log_dir is not None:
            os.makedirs(log_dir, exist_ok=True)
            self._monitors = [os.path.join(log_dir, "{}.monitor.csv".format(first_env_id + i)) for i in range(self.num_envs)]
            header = "#%s\nr,l,t\n" % json.dumps({"t_start": self._t_start, "env_id": env_id})
            for path in self._monitors:
                with open(path, "wt") as f:
                    f.write(header)
            # buffered rows must not die with a process that never calls close() (rl_baselines.train leaves that to interpreter exit)
            import atexit
            import weakref
            ref = weakref.ref(self)
            atexit.register(lambda: ref() is not None and ref()._monitors is not None and ref()._mon_rows and ref()._flush_monitors())

    # -- VecEnv API ----------------------------------------------------------------
    def _encode(self):
        """states of the frames the stepper just rendered (its stream) -> float32 numpy [N][state_dim]"""
        import torch
        if getattr(self._enc, "hip", None) is not None:
            st = self._enc.getStates(self._t["images"], stream=self._h.stream())     # same stream: ordered behind the rasteriser
            self._h.sync()
        else:
            self._h.sync()
            st = self._enc.getStates(self._t["images"])
            torch.cuda.synchronize()
        return st.to("cpu").numpy()

    def reset(self):
        # stable_baselines.bench.Monitor(allow_early_resets=False) refuses a reset() in the middle of an episode
        # (environments/utils.py:54, rl_baselines/utils.py:194): here that is a reset while some env's running episode has steps
        # (checked first so that allow_early_resets costs no device read)
        if not self.allow_early_resets and self._was_reset and (self._h.get_state(_lib.F_EP_LENGTH) > 0).any():
            raise RuntimeError("Tried to reset an environment before done. If you want to allow early resets, "
                               "wrap your env with Monitor(env, path, allow_early_resets=True)")
        self._was_reset = True
        if self._monitors is not None and self._mon_rows:
            self._flush_monitors()
        if self._enc is None:
            return self._h.reset()
        self._h.reset(obs_out=self._t["images"].data_ptr())
        return self._encode()

    def step_async(self, actions):
        f = self._fast
        if self.cfg.is_discrete:
            if isinstance(actions, np.ndarray) and actions.dtype != object:      # fast path: no `None` entries possible
                if f is not None:
                    self._actions = f["act"]
                    np.copyto(self._actions, actions.reshape(self.num_envs), casting="unsafe")
                else:
                    self._actions = np.ascontiguousarray(actions, dtype=np.int32).reshape(self.num_envs)
            else:
                self._actions = np.array([-1 if a is None else int(a) for a in actions], dtype=np.int32)
                if f is not None:
                    f["act"][:] = self._actions
                    self._actions = f["act"]
            if self._actions.size and (self._actions.min() < -1 or self._actions.max() >= self._h.num_actions):
                # the reference indexes a per-action list (`[-dv, dv, 0, 0, 0, 0][action]`): IndexError in the worker
                raise IndexError("discrete action out of range [0, {}) (None/-1 = no-op)".format(self._h.num_actions))
        else:
            # (a numeric ndarray cannot hold None: the per-element scan below would cost a 4096-iteration Python loop per step)
            if not (isinstance(actions, np.ndarray) and actions.dtype != object) and any(a is None for a in actions):
                raise NotImplementedError("None actions need a discrete action space")
            self._actions = np.asarray(actions, dtype=np.float32).reshape(self.num_envs, self._h.action_dim)
            if f is not None:
                f["act"][...] = self._actions
                self._actions = f["act"]

    def _flush_monitors(self):
        """Monitor rows are buffered per env and appended in one write per file: at 4096 envs some episode ends in every step, and an
        open / write / close per episode (15 us each) would cost more than the step.  Batches of up to 64 envs write through; larger ones
        flush every 512 rows, every second, on reset()
        and on close() — the reference's Monitor flushes per episode; a reader of the CSV files sees rows at most a second late."""
        for i, rows in self._mon_rows.items():
            fd = os.open(self._monitors[i], os.O_WRONLY | os.O_APPEND)
            try:
                os.write(fd, "".join(rows).encode())
            finally:
                os.close(fd)
        self._mon_rows, self._mon_count, self._mon_flushed = {}, 0, time.time()

    def step_wait(self):
        f = self._fast
        if f is not None:
            rc = f["call"]()
            if rc:
                self._h._check(rc, "srlhip_step")
            # fresh arrays per step, like SubprocVecEnv's np.stack (callers keep references across steps)
            obs, rew, done = f["obs"].copy(), f["rew"].copy(), f["done"]
        elif self._enc is None:
            obs, rew, done = self._h.step(self._actions)
        else:
            import torch
            t = self._t
            t["act"].copy_(torch.from_numpy(self._actions))
            torch.cuda.current_stream().synchronize()                    # the actions must be in HBM before the stepper's stream reads them
            self._h.step(t["act"].data_ptr(), out=(t["images"].data_ptr(), t["rew"].data_ptr(), t["done"].data_ptr()))
            obs = self._encode()                                         # syncs the stepper's stream
            rew, done = t["rew"].to("cpu").numpy(), t["done"].to("cpu").numpy()
        dones = done.astype(bool)
        infos = self._quiet_infos
        if dones.any():
            idx = np.flatnonzero(dones)
            if f is not None:
                ret, length = f["ret"][idx], f["len"][idx]          # mapped record planes: final once the step call has returned
                self._n_finished[idx] += 1
            else:
                ret, length, fin = self._h.episode_stats()
                ret, length, self._n_finished = ret[idx], length[idx], fin
            now = time.time()
            t = round(now - self._t_start, 6)
            infos = list(infos)
            for k, i in enumerate(idx.tolist()):
                ep = {"r": round(float(ret[k]), 6), "l": int(length[k]), "t": t}
                infos[i] = {"episode": ep}
                if self._monitors is not None:
                    self._mon_rows.setdefault(i, []).append("{},{},{}\n".format(ep["r"], ep["l"], ep["t"]))
            if self._monitors is not None:
                self._mon_count += len(idx)
                if self.num_envs <= 64 or self._mon_count >= 512 or now - self._mon_flushed >= 1.0:
                    self._flush_monitors()
        return obs, rew, dones, infos

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def rollout(self, n_steps, actions=None):
        """Fused device-side rollout (no per-step host round trip): dict of [T][N] planes."""
        if self._enc is not None:
            raise NotImplementedError("fused rollouts with a learned SRL encoder: use srlhip.pixel_env.PixelStateVecEnv")
        return self._h.rollout(n_steps, actions=actions)

    def get_images(self):
        if self._enc is not None:
            self._h.render(out=self._t["images"].data_ptr())
            self._h.sync()
            return list(self._t["images"].to("cpu").numpy())
        return list(self._h.render())

    def render(self, mode="human"):
        return np.array([])

    def seed(self, seed):
        self._h.seed(int(seed) + self.cfg.first_env_id + np.arange(self.num_envs, dtype=np.int64))

    def episode_returns(self):
        """Per-env return/length of the last finished episode and the number of finished episodes."""
        return self._h.episode_stats()

    def close(self):
        if self._monitors is not None and self._mon_rows:
            self._flush_monitors()
        self._monitors = None
        self._fast = None                                # numpy views into the handle's mapped memory die with it
        if self._h is not None:
            self._h.close()
            self._h = None

    @property
    def unwrapped(self):
        return self
