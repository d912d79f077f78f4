"""Per-step API rates (run on the GPU box): HipVecEnv.step with host numpy in/out, 4096 envs."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "robotics-rl-srl_amd"))
import numpy as np
import torch  # noqa: F401
from srlhip.vec_env import HipVecEnv
for env_id, nact in (("MobileRobotGymEnv-v0", 4), ("KukaButtonGymEnv-v0", 6)):
    venv = HipVecEnv(env_id, 4096, seed=0, env_kwargs={"srl_model": "ground_truth"})
    venv.reset()
    acts = np.random.RandomState(0).randint(nact, size=(400, 4096))
    for t in range(50): venv.step(acts[t])
    t0 = time.perf_counter()
    for t in range(50, 350): venv.step(acts[t])
    dt = (time.perf_counter() - t0) / 300
    print("{}: {:.1f} us per 4096-env VecEnv.step -> {:.2e} env-steps/s".format(env_id, dt * 1e6, 4096 / dt))
    a32 = acts.astype(np.int32)
    out = venv._h.step(a32[0])
    t0 = time.perf_counter()
    for t in range(50, 350): venv._h.step(a32[t], out=out)
    dt = (time.perf_counter() - t0) / 300
    print("{}: {:.1f} us per 4096-env srlhip_step (ctypes, reused numpy buffers) -> {:.2e} env-steps/s".format(env_id, dt * 1e6, 4096 / dt))
    venv.close()
