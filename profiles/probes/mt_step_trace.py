import os, sys
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "robotics-rl-srl_amd"))
import numpy as np, torch
from srlhip import _lib
cfg = _lib.default_config(_lib.ENV_KUKA_BUTTON)
cfg.num_envs, cfg.rng_mode, cfg.auto_reset, cfg.io_device = 4096, _lib.RNG_MT19937, 1, 1
h = _lib.Handle(cfg)
acts = torch.from_numpy(np.random.RandomState(0).randint(6, size=(700, 4096)).astype(np.int32)).cuda()
o = torch.zeros((4096, 3), device="cuda"); r = torch.zeros(4096, device="cuda"); d = torch.zeros(4096, dtype=torch.uint8, device="cuda")
h.reset(obs_out=o.data_ptr()); h.sync()
for t in range(700): h.step(acts[t].data_ptr(), out=(o.data_ptr(), r.data_ptr(), d.data_ptr()))
h.sync()
