import sys, os
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "robotics-rl-srl_amd"))
import torch
from srlhip import _lib
for kind, hw in ((4, 64), (4, 224), (0, 64)):
    cfg = _lib.default_config(kind)
    cfg.num_envs, cfg.seed0, cfg.obs_mode, cfg.img_h, cfg.img_w, cfg.io_device = 4096 if hw == 64 else 512, 0, _lib.OBS_RAW_PIXELS, hw, hw, 1
    h = _lib.Handle(cfg)
    img = torch.zeros((cfg.num_envs, hw, hw, 3), dtype=torch.uint8, device="cuda")
    h.reset(obs_out=img.data_ptr()); h.sync()
    for _ in range(3): h.render(out=img.data_ptr())
    h.sync(); h.timing_begin()
    for _ in range(50): h.render(out=img.data_ptr())
    print("kind", kind, "hw", hw, "envs", cfg.num_envs, "raster ms:", h.timing_end() / 50)
    h.close()
