"""raw_pixels -> SRL state pipeline on one device (BASELINE configs 4-5): the stepper and the tile
rasteriser write uint8 images straight into a torch-owned HBM buffer (io_device = 1, zero host copies),
and the SRL encoder runs one batched forward on it — instead of the reference's one-image-at-a-time
encoder server behind multiprocessing queues (rl_baselines/utils.py:162-191)."""
import numpy as np
import torch

from . import _lib
from .envs import ENV_CLASSES


class PixelStateVecEnv(object):
    def __init__(self, env_id, num_envs, encoder, seed=0, img_shape=(64, 64), device_id=0, first_env_id=0,
                 rng_mode=_lib.RNG_PHILOX, env_kwargs=None):
        kw = dict(env_kwargs or {})
        cfg = _lib.default_config(ENV_CLASSES[env_id].ENV_KIND)
        cfg.num_envs, cfg.device_id, cfg.first_env_id, cfg.seed0 = num_envs, device_id, first_env_id, seed
        cfg.random_target, cfg.multi_view = int(kw.get("random_target", False)), int(bool(kw.get("multi_view", False)) or bool(kw.get("fpv", False)))
        cfg.obs_mode, cfg.img_h, cfg.img_w = _lib.OBS_RAW_PIXELS, img_shape[0], img_shape[1]
        cfg.rng_mode, cfg.auto_reset, cfg.io_device = rng_mode, 1, 1
        self.h = _lib.Handle(cfg)
        self.encoder, self.num_envs = encoder, num_envs
        self.device = torch.device("cuda", device_id)
        ch = 6 if cfg.multi_view else 3
        self.images = torch.zeros((num_envs, img_shape[0], img_shape[1], ch), dtype=torch.uint8, device=self.device)
        self.rewards = torch.zeros((num_envs,), dtype=torch.float32, device=self.device)
        self.dones = torch.zeros((num_envs,), dtype=torch.uint8, device=self.device)
        self.actions = torch.zeros((num_envs,), dtype=torch.int32, device=self.device)
        self.states = torch.zeros((num_envs, encoder.state_dim), dtype=torch.float32, device=self.device)
        self._stream_ptr = self.h.stream()
        self._stream = torch.cuda.ExternalStream(self._stream_ptr, device=self.device)

    def _encode(self):
        if self.encoder.hip is not None:
            # fused HIP encoder: enqueue on the stepper's own stream right behind the rasteriser (no host sync), then
            # order torch's current stream behind it so the caller can consume states / rewards / dones as usual
            states = self.encoder.getStates(self.images, stream=self._stream_ptr, out=self.states)
            torch.cuda.current_stream(self.device).wait_stream(self._stream)
            return states
        self.h.sync()                                  # images were written on the stepper's stream
        return self.encoder.getStates(self.images)

    def reset(self):
        self._stream.wait_stream(torch.cuda.current_stream(self.device))
        self.h.reset(obs_out=self.images.data_ptr())
        return self._encode()

    def step(self, actions=None):
        """actions: int32 device tensor [N] (None -> device-sampled random agent).  -> states, rewards, dones
        (device tensors, overwritten by the next call)."""
        # the stepper's stream overwrites images / states / rewards / dones: wait for their readers (and for `actions`)
        self._stream.wait_stream(torch.cuda.current_stream(self.device))
        if actions is None:
            self.h.rollout(1, out=(self.images.data_ptr(), self.rewards.data_ptr(), self.dones.data_ptr(), self.actions.data_ptr()))
        else:
            self.h.step(actions.data_ptr(), out=(self.images.data_ptr(), self.rewards.data_ptr(), self.dones.data_ptr()))
        return self._encode(), self.rewards, self.dones

    def close(self):
        self.h.close()
