#!/usr/bin/env python
"""mobile_chain_probe.py — the MobileRobot episode-parallel rollout with CALLER-SUPPLIED actions (no sampler workgroups beside the
segment waves) vs the synthetic agent (sampler workgroups of the next action plane inside the same launch): what the 251-step
recurrence costs alone.  HIP events on the stepper's own stream (srlhip_timing_*).  Run on the GPU box from the repo root."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "robotics-rl-srl_amd"))
import torch
from srlhip import _lib

n, T, K = 4096, 2048, 50
dev = torch.device("cuda:0")
for mode in ("given", "agent"):
    cfg = _lib.default_config(_lib.ENV_MOBILE)
    cfg.num_envs, cfg.rng_mode, cfg.auto_reset, cfg.io_device = n, _lib.RNG_PHILOX, 1, 1
    h = _lib.Handle(cfg)
    obs0 = torch.zeros((n, 2), dtype=torch.float32, device=dev)
    h.reset(obs_out=obs0.data_ptr())
    bufs = (torch.zeros((T, n, 2), dtype=torch.float32, device=dev), torch.zeros((T, n), dtype=torch.float32, device=dev),
            torch.zeros((T, n), dtype=torch.uint8, device=dev), torch.zeros((T, n), dtype=torch.int32, device=dev))
    out = (bufs[0].data_ptr(), bufs[1].data_ptr(), bufs[2].data_ptr(), bufs[3].data_ptr() if mode == "agent" else None)
    plane = torch.randint(0, 4, (T, n), dtype=torch.int32, device=dev)
    actions = plane.data_ptr() if mode == "given" else None
    torch.cuda.synchronize()
    for _ in range(5):
        h.rollout(T, actions=actions, out=out)
    torch.cuda.synchronize()
    h.timing_begin()
    for _ in range(K):
        h.rollout(T, actions=actions, out=out)
    ms = h.timing_end()
    print("{}: {:.4f} ms per {}-env x {}-step rollout = {:.3e} env-steps/s".format(mode, ms / K, n, T, n * T * K / ms * 1e3))
    h.close()
