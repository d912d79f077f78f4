"""Per-step cost by RNG mode and io mode (run on the GPU box); SRLHIP_ZERO_COPY=0 selects the bounce-buffer path."""
import os, sys, time
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "robotics-rl-srl_amd"))
import numpy as np, torch
from srlhip import _lib
# (the first configuration also pays clock ramp-up / lazy initialisation: listed twice, read the second)
for kind, nact in ((_lib.ENV_KUKA_BUTTON, 6), (_lib.ENV_KUKA_BUTTON, 6), (_lib.ENV_MOBILE, 4)):
    for mode, name in ((_lib.RNG_PHILOX, "philox"), (_lib.RNG_MT19937, "mt19937")):
        for io in (0, 1):
            cfg = _lib.default_config(kind)
            cfg.num_envs, cfg.rng_mode, cfg.auto_reset, cfg.io_device = 4096, mode, 1, io
            h = _lib.Handle(cfg)
            acts = np.random.RandomState(0).randint(nact, size=(400, 4096)).astype(np.int32)
            if io:
                a = torch.from_numpy(acts).cuda(); o = torch.zeros((4096, h.obs_dim), device="cuda"); r = torch.zeros(4096, device="cuda"); d = torch.zeros(4096, dtype=torch.uint8, device="cuda")
                h.reset(obs_out=o.data_ptr()); h.sync()
                f = lambda t: h.step(a[t].data_ptr(), out=(o.data_ptr(), r.data_ptr(), d.data_ptr()))
            else:
                h.reset(); out = h.step(acts[0])
                f = lambda t: h.step(acts[t], out=out)
            for t in range(30): f(t)
            h.sync(); t0 = time.perf_counter()
            for t in range(30, 330): f(t)
            h.sync(); dt = (time.perf_counter() - t0) / 300
            print("kind", kind, name, "io_device", io, "%.1f us/step" % (dt * 1e6))
            h.close()
