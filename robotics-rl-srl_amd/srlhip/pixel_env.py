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
                 rng_mode=_lib.RNG_PHILOX, env_kwargs=None, use_graph=False):
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
        # use_graph: replay stepper + rasteriser + encoder of a step from a HIP graph (srlhip_graph_*), keyed by the action
        # source.  Off by default: at 4096 envs the step is GPU-bound and the host already runs ahead of it (697 us eager
        # vs 702 us replayed, profiles/pixel_step_microbench.py); it pays when the host thread is the bottleneck.
        self.use_graph, self._graphs, self._warm = bool(use_graph), {}, {}
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
        fused = self.encoder.hip is not None
        key = "sampled" if actions is None else ("given", actions.data_ptr())
        g = self._graphs.get(key) if (fused and self.use_graph) else None
        if g is not None:
            # stepper + rasteriser + encoder of one VecEnv step replayed from a HIP graph: one launch instead of three
            self.h.graph_launch(g)
            torch.cuda.current_stream(self.device).wait_stream(self._stream)
            return self.states, self.rewards, self.dones
        # the first call runs eagerly (lazy allocations); at most a handful of distinct action buffers get their own graph
        capture = fused and self.use_graph and self._warm.get(key, 0) >= 1 and len(self._graphs) < 8
        if capture:
            self.h.graph_begin()
        if actions is None:
            self.h.rollout(1, out=(self.images.data_ptr(), self.rewards.data_ptr(), self.dones.data_ptr(), self.actions.data_ptr()))
        else:
            self.h.step(actions.data_ptr(), out=(self.images.data_ptr(), self.rewards.data_ptr(), self.dones.data_ptr()))
        if capture:
            self.encoder.getStates(self.images, stream=self._stream_ptr, out=self.states)
            self._graphs[key] = self.h.graph_end()
            self.h.graph_launch(self._graphs[key])                              # the captured step has not run yet
            torch.cuda.current_stream(self.device).wait_stream(self._stream)
            return self.states, self.rewards, self.dones
        self._warm[key] = self._warm.get(key, 0) + 1
        return self._encode(), self.rewards, self.dones

    def close(self):
        for g in self._graphs.values():
            self.h.graph_destroy(g)
        self._graphs = {}
        self.h.close()
