"""How much of HipVecEnv.step is Python: the two foreign calls (srlhip_step_async / srlhip_step_wait) timed inside step(), 4096 envs, steady
state, launching and persistent paths."""
import sys, time
import numpy as np
sys.path.insert(0, "robotics-rl-srl_amd")
from srlhip.vec_env import HipVecEnv
n = 4096
for persistent in (False, True):
    env = HipVecEnv("KukaButtonGymEnv-v0", n, seed=0, env_kwargs={"srl_model": "ground_truth"}, persistent=persistent)
    env.reset()
    acts = np.random.RandomState(0).randint(6, size=(1500, n))
    for t in range(1100): env.step(acts[t])
    sh = env._shards[0]
    go, collect = sh.go, sh.collect
    tg, tc = [], []
    def go2():
        t0 = time.perf_counter(); r = go(); tg.append(time.perf_counter() - t0); return r
    def collect2():
        t0 = time.perf_counter(); r = collect(); tc.append(time.perf_counter() - t0); return r
    sh.go, sh.collect = go2, collect2
    ts = []
    for t in range(1100, 1500):
        t0 = time.perf_counter(); env.step(acts[t]); ts.append(time.perf_counter() - t0)
    ts, tg, tc = np.array(ts) * 1e6, np.array(tg) * 1e6, np.array(tc) * 1e6
    print("%s: step median %.1f us = srlhip_step_async %.1f + srlhip_step_wait %.1f + Python %.1f" % (
        "persistent" if persistent else "launching ", np.median(ts), np.median(tg), np.median(tc), np.median(ts - tg - tc)))
    env.close()
