"""End-to-end latency of the per-step API: HipVecEnv.step (numpy in / numpy out, Monitor bookkeeping) and the raw srlhip_step behind
it, KukaButtonGymEnv ground truth, 16 ... 4096 envs.  Run on the GPU box from the repo root."""
import os, sys, time
import numpy as np
REPO = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(REPO, "robotics-rl-srl_amd")); sys.path.insert(0, REPO)
from srlhip.vec_env import HipVecEnv
from srlhip import _lib
for n in (16, 64, 256, 4096):
    env = HipVecEnv("KukaButtonGymEnv-v0", n, seed=0, env_kwargs={"srl_model": "ground_truth"})
    env.reset()
    acts = np.random.RandomState(0).randint(6, size=(300, n))
    for t in range(50): env.step(acts[t])
    ts = []
    for t in range(50, 300):
        t0 = time.perf_counter(); env.step(acts[t]); ts.append(time.perf_counter() - t0)
    ts = np.array(ts) * 1e6
    print("HipVecEnv n=%d: mean %.1f us per step, median %.1f, p90 %.1f, max %.1f (%.3g env-steps/s)" % (n, ts.mean(), np.median(ts), np.percentile(ts, 90), ts.max(), n / ts.mean() * 1e6), flush=True)
    env.close()
    cfg = _lib.default_config(_lib.ENV_KUKA_BUTTON)
    cfg.num_envs, cfg.seed0 = n, 0
    h = _lib.Handle(cfg)
    h.reset()
    a = acts[:, :].astype(np.int32)
    for t in range(50): h.step(a[t])
    ts = []
    for t in range(50, 300):
        t0 = time.perf_counter(); h.step(a[t]); ts.append(time.perf_counter() - t0)
    ts = np.array(ts) * 1e6
    print("   raw srlhip_step (host numpy io) n=%d: mean %.1f us, median %.1f, p90 %.1f, max %.1f" % (n, ts.mean(), np.median(ts), np.percentile(ts, 90), ts.max()), flush=True)
    h.close()
