"""End-to-end latency of the per-step API in STEADY STATE (every env past its first episode, so that episodes end in nearly every
step of a 4096-env batch): HipVecEnv.step (numpy in / numpy out, Monitor bookkeeping, info['episode']) and the raw srlhip_step behind
it, KukaButtonGymEnv ground truth, 16 ... 4096 envs.  Run on the GPU box from the repo root; optional argv[1] = warm-up steps."""
import os, sys, time
import numpy as np
REPO = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(REPO, "robotics-rl-srl_amd")); sys.path.insert(0, REPO)
import torch  # noqa: F401  (first: __graft_entry__.build())
from srlhip.vec_env import HipVecEnv
from srlhip import _lib
WARM = int(sys.argv[1]) if len(sys.argv) > 1 else 1100
TIMED = 400
for n in (16, 256, 4096):
    env = HipVecEnv("KukaButtonGymEnv-v0", n, seed=0, env_kwargs={"srl_model": "ground_truth"})
    env.reset()
    acts = np.random.RandomState(0).randint(6, size=(WARM + TIMED, n))
    for t in range(WARM): env.step(acts[t])
    ts, nd = [], 0
    for t in range(WARM, WARM + TIMED):
        t0 = time.perf_counter(); o, r, d, info = env.step(acts[t]); ts.append(time.perf_counter() - t0); nd += int(d.any())
    ts = np.array(ts) * 1e6
    print("HipVecEnv n=%d: mean %.1f us per step, median %.1f, p90 %.1f, max %.1f (%.3g env-steps/s); an episode ended in %d of %d steps" % (
        n, ts.mean(), np.median(ts), np.percentile(ts, 90), ts.max(), n / ts.mean() * 1e6, nd, TIMED), flush=True)
    env.close()
    cfg = _lib.default_config(_lib.ENV_KUKA_BUTTON)
    cfg.num_envs, cfg.seed0 = n, 0
    h = _lib.Handle(cfg)
    h.reset()
    a = acts.astype(np.int32)
    out = (h.new_obs(), np.zeros(n, np.float32), np.zeros(n, np.uint8))
    for t in range(WARM): h.step(a[t], out=out)
    ts = []
    for t in range(WARM, WARM + TIMED):
        t0 = time.perf_counter(); h.step(a[t], out=out); ts.append(time.perf_counter() - t0)
    ts = np.array(ts) * 1e6
    print("   raw srlhip_step (host numpy io, preallocated outputs) n=%d: mean %.1f us, median %.1f, p90 %.1f, max %.1f" % (n, ts.mean(), np.median(ts), np.percentile(ts, 90), ts.max()), flush=True)
    ts = []
    for t in range(50):
        t0 = time.perf_counter(); h.episode_stats(); ts.append(time.perf_counter() - t0)
    print("   srlhip_episode_stats n=%d: median %.1f us" % (n, np.median(np.array(ts) * 1e6)), flush=True)
    h.close()
# round 6: the launch / collect split of a step (step_async enqueues every shard's kernel, step_wait collects) and the SAME 4096 envs
# as 2 and 4 shards of one process (HipVecEnv(device_ids=[...]); here every shard lives on device 0: the host-side cost of sharding
# and the concurrency of the shards' launches, not a multi-GPU number)
for ids in ([0], [0, 0], [0, 0, 0, 0]):
    n = 4096
    env = HipVecEnv("KukaButtonGymEnv-v0", n, seed=0, env_kwargs={"srl_model": "ground_truth"}, device_ids=ids)
    env.reset()
    acts = np.random.RandomState(0).randint(6, size=(WARM + TIMED, n))
    for t in range(WARM): env.step(acts[t])
    ta, tw = [], []
    for t in range(WARM, WARM + TIMED):
        t0 = time.perf_counter(); env.step_async(acts[t]); t1 = time.perf_counter(); env.step_wait(); t2 = time.perf_counter()
        ta.append(t1 - t0); tw.append(t2 - t1)
    ta, tw = np.array(ta) * 1e6, np.array(tw) * 1e6
    print("HipVecEnv n=%d as %d shard(s) on device 0: step median %.1f us = step_async %.1f (launches only) + step_wait %.1f" % (
        n, len(ids), np.median(ta + tw), np.median(ta), np.median(tw)), flush=True)
    env.close()
# round 6: persistent stepping (srlhip_set_persistent / HipVecEnv(persistent=True)): the same loops without a launch per step
for n in (16, 256, 4096):
    env = HipVecEnv("KukaButtonGymEnv-v0", n, seed=0, env_kwargs={"srl_model": "ground_truth"}, persistent=True)
    env.reset()
    acts = np.random.RandomState(0).randint(6, size=(WARM + TIMED, n))
    for t in range(WARM): env.step(acts[t])
    ts = []
    for t in range(WARM, WARM + TIMED):
        t0 = time.perf_counter(); env.step(acts[t]); ts.append(time.perf_counter() - t0)
    ts = np.array(ts) * 1e6
    print("PERSISTENT HipVecEnv n=%d: mean %.1f us per step, median %.1f, p90 %.1f, max %.1f (%.3g env-steps/s)" % (n, ts.mean(), np.median(ts), np.percentile(ts, 90), ts.max(), n / ts.mean() * 1e6), flush=True)
    env.close()
    cfg = _lib.default_config(_lib.ENV_KUKA_BUTTON)
    cfg.num_envs, cfg.seed0 = n, 0
    h = _lib.Handle(cfg)
    h.reset()
    h.set_persistent(True)
    a = acts.astype(np.int32)
    out = (h.new_obs(), np.zeros(n, np.float32), np.zeros(n, np.uint8))
    for t in range(WARM): h.step(a[t], out=out)
    ts = []
    for t in range(WARM, WARM + TIMED):
        t0 = time.perf_counter(); h.step(a[t], out=out); ts.append(time.perf_counter() - t0)
    ts = np.array(ts) * 1e6
    print("   PERSISTENT raw srlhip_step n=%d: mean %.1f us, median %.1f, p90 %.1f, max %.1f" % (n, ts.mean(), np.median(ts), np.percentile(ts, 90), ts.max()), flush=True)
    h.close()
