"""Per-launch cost of the Kuka rollout kernel: ms per launch for T = 1, 2, 4, 16, 64 steps (4096 envs, device buffers, Philox)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "robotics-rl-srl_amd")); sys.path.insert(0, REPO)
import torch
from srlhip import _lib
for model in (_lib.KUKA_MODEL_FULL, _lib.KUKA_MODEL_LUMPED):
    cfg = _lib.default_config(_lib.ENV_KUKA_BUTTON)
    cfg.num_envs, cfg.rng_mode, cfg.auto_reset, cfg.io_device, cfg.kuka_model = 4096, _lib.RNG_PHILOX, 1, 1, model
    h = _lib.Handle(cfg)
    dev = torch.device("cuda", 0)
    T = 64
    obs = torch.zeros((T, 4096, 3), device=dev); rew = torch.zeros((T, 4096), device=dev); done = torch.zeros((T, 4096), dtype=torch.uint8, device=dev); act = torch.zeros((T, 4096), dtype=torch.int32, device=dev)
    o0 = torch.zeros((4096, 3), device=dev)
    h.reset(obs_out=o0.data_ptr()); h.sync()
    out = (obs.data_ptr(), rew.data_ptr(), done.data_ptr(), act.data_ptr())
    for t in (1, 2, 4, 16, 64):
        for _ in range(5): h.rollout(t, out=out)
        h.sync(); h.timing_begin()
        for _ in range(50): h.rollout(t, out=out)
        ms = h.timing_end() / 50
        print(h.kuka_kernel(), "T=%d: %.4f ms per launch, %.2f us per step" % (t, ms, ms * 1e3 / t))
    h.close()
