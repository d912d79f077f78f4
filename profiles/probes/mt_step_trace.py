"""Per-step launches in the reference-exact RNG mode (MT19937 generators resident in HBM), 4096 envs, device pointers:
run under `rocprofv3 --kernel-trace` to see the per-launch kernel durations.  argv[1]: kuka | mobile"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "robotics-rl-srl_amd"))
import numpy as np, torch
from srlhip import _lib
kind = _lib.ENV_MOBILE if (len(sys.argv) > 1 and sys.argv[1] == "mobile") else _lib.ENV_KUKA_BUTTON
mode = _lib.RNG_PHILOX if (len(sys.argv) > 2 and sys.argv[2] == "philox") else _lib.RNG_MT19937
cfg = _lib.default_config(kind)
cfg.num_envs, cfg.rng_mode, cfg.auto_reset, cfg.io_device = 4096, mode, 1, 1
h = _lib.Handle(cfg)
acts = torch.from_numpy(np.random.RandomState(0).randint(4, size=(700, 4096)).astype(np.int32)).cuda()
o = torch.zeros((4096, 3), device="cuda"); r = torch.zeros(4096, device="cuda"); d = torch.zeros(4096, dtype=torch.uint8, device="cuda")
h.reset(obs_out=o.data_ptr()); h.sync()
for t in range(700): h.step(acts[t].data_ptr(), out=(o.data_ptr(), r.data_ptr(), d.data_ptr()))
h.sync()
