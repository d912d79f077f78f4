#!/usr/bin/env python
"""Parity soak of the headline configuration: HIP stepper (through the C-ABI) vs the oracle over several seed bases at the BASELINE
size, every plane of every step compared — a larger sample behind "reward / done bit for bit, joints 1e-11" than the test suite's
4.1e6 env-steps (tests/test_gpu_kuka.py::test_fused_rollout_4096_envs, whose construction this follows).
Runs on the GPU box (the oracle leg uses the host's cores: ~2 minutes per base):
    python profiles/probes/parity_soak.py [bases=3] [n=4096] [T=1001] [mode=given] > gpurun_out/r05_parity_soak.json
    python profiles/probes/parity_soak.py 4 4096 2048 philox > gpurun_out/r05_parity_soak_philox.json
mode philox = the launch bench.py times: Philox env streams, actions sampled on the device by the synthetic agent (compared too).
mode perstep = the per-step API (srlhip_step, host numpy buffers: what HipVecEnv / rl_baselines.train drive), one launch per step
(round 6: reported by the kernel's early completion signal); mode persistent = the same calls under srlhip_set_persistent.
mode continuous | joints = KukaButton with 3-D Cartesian / 7-D joint-space Box actions (uniform in [-1, 1]).
mode moving | two | rand = the env variants at the BASELINE size (KukaMovingButton with shape_reward, Kuka2Button, KukaRandButton with
random_target; MT19937 streams, given actions): the test suite compares them at 128 envs.
Per base b: seeds 100000 b + (0..n-1) (numpy MT19937 streams, the reference's seeding), actions uniform over the 6 discrete actions
with 25 % extra 'down' (episodes end by contact well before 1001 steps: every env crosses >= 1 auto-reset), one fused launch.
Reported: mismatching reward / done entries, max |obs - obs_oracle| (float32 planes), max |q - q_oracle| of the final state, the
smallest |value - threshold| behind any contact / distance flag (oracle margin probe), episodes finished.
TEST INFRASTRUCTURE (drives the oracle next to the product)."""
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(REPO, "robotics-rl-srl_amd"), REPO):
    sys.path.insert(0, p)
import torch  # noqa: F401,E402  (before libsrlhip.so: __graft_entry__.build())
from oracle import kuka_clib  # noqa: E402
from srlhip import _lib  # noqa: E402

bases = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
T = int(sys.argv[3]) if len(sys.argv) > 3 else 1001
mode = sys.argv[4] if len(sys.argv) > 4 else "given"
philox = mode == "philox"
KIND = {"moving": _lib.ENV_KUKA_MOVING, "two": _lib.ENV_KUKA_2BUTTON, "rand": _lib.ENV_KUKA_RAND}.get(mode, _lib.ENV_KUKA_BUTTON)
ORA_KW = {"continuous": dict(is_discrete=False), "joints": dict(is_discrete=False, action_joints=True), "moving": dict(shape_reward=True), "two": dict(force_down=False, max_distance=2.0), "rand": dict(random_target=True)}.get(mode, {})
kuka_clib.set_full(True)
res = {"n": n, "T": T, "mode": "philox streams, device-sampled actions" if philox else "MT19937 streams, given actions" + ("" if mode == "given" else ", env variant " + mode), "bases": [], "env_steps": 0, "reward_mismatches": 0, "done_mismatches": 0}
for b in range(1, bases + 1):
    seed0 = 100000 * b
    rs = np.random.RandomState(1000 + b)
    actions = rs.randint(6, size=(T, n)).astype(np.int32)
    actions[rs.rand(T, n) < 0.25] = 4
    cfg = _lib.default_config(KIND)
    cfg.num_envs, cfg.rng_mode, cfg.auto_reset, cfg.seed0 = n, _lib.RNG_PHILOX if philox else _lib.RNG_MT19937, 1, seed0
    if mode == "moving":
        cfg.shape_reward = 1
    if mode in ("continuous", "joints"):
        cfg.is_discrete, cfg.action_joints = 0, int(mode == "joints")
        actions = rs.uniform(-1, 1, size=(T, n, 7 if mode == "joints" else 3)).astype(np.float32)
    if mode == "rand":
        cfg.random_target = 1
    h = _lib.Handle(cfg)
    obs0 = h.reset()
    if mode in ("perstep", "persistent"):
        if mode == "persistent":
            h.set_persistent(True)                       # round 6: no launch per step (srlhip_set_persistent)
        od = obs0.shape[1]
        out = {"obs": np.zeros((T, n, od), np.float32), "reward": np.zeros((T, n), np.float32), "done": np.zeros((T, n), np.uint8), "actions": actions}
        for t in range(T):
            h.step(actions[t], out=(out["obs"][t], out["reward"][t], out["done"][t]))
    else:
        out = h.rollout(T) if philox else h.rollout(T, actions=actions)
    kuka_clib.margins_reset()
    t0 = time.time()
    if mode == "moving":
        kuka_clib.set_moving(True)
    if mode in ("two", "rand"):
        kuka_clib.set_variant(kuka_clib.VARIANT_TWO if mode == "two" else kuka_clib.VARIANT_RAND)
    try:
        if philox:
            ora = kuka_clib.rollout(seed0 + np.arange(n), T, actions=None, rng_mode=kuka_clib.RNG_PHILOX, trace=False)
        else:
            ora = kuka_clib.rollout(seed0 + np.arange(n), T, actions=actions, trace=False, **ORA_KW)
    finally:
        kuka_clib.set_moving(False)
        kuka_clib.set_variant(kuka_clib.VARIANT_BUTTON)
    t_ora = time.time() - t0
    m = kuka_clib.margins()
    f = ora["final_state"]
    ret, length, fin = h.episode_stats()
    st = {"seed0": seed0, "action_mismatches": int((ora["actions"] != out["actions"]).sum()) if philox else 0,
          "reward_mismatches": int((ora["reward"] != out["reward"]).sum()), "max_reward_diff": float(np.abs(ora["reward"] - out["reward"]).max()), "done_mismatches": int((ora["done"] != out["done"]).sum()),
          "max_obs_diff": float(max(np.abs(ora["obs"] - out["obs"]).max(), np.abs(ora["obs0"] - obs0).max())),
          "max_final_q_diff": float(np.abs(h.get_state(_lib.F_KUKA_Q).T - f[:, 0:7]).max()),
          "final_q_diff_percentiles_50_90_99_99.9_100": np.percentile(np.abs(h.get_state(_lib.F_KUKA_Q).T - f[:, 0:7]).max(axis=1), [50, 90, 99, 99.9, 100]).tolist(),
          "worst_envs": np.argsort(np.abs(h.get_state(_lib.F_KUKA_Q).T - f[:, 0:7]).max(axis=1))[-4:].tolist(),
          "obs_diff_percentiles_per_env_50_99_100": np.percentile(np.abs(ora["obs"] - out["obs"]).max(axis=(0, 2)), [50, 99, 100]).tolist(),
          "episode_counts_equal": bool(np.array_equal(fin, ora["ep_stats"][:, 2].astype(np.int32)) and np.array_equal(length, ora["ep_stats"][:, 1].astype(np.int32))),
          "episodes_finished": int(fin.sum()), "min_episodes_per_env": int(fin.min()),
          "ik_flagged_env_steps": int((h.get_state(_lib.F_KUKA_IK_CROSSED) >> 1).sum()),
          "flag_margins": {k: float(v) for k, v in m.items()}, "oracle_seconds": round(t_ora, 1)}
    h.close()
    res["bases"].append(st)
    res["env_steps"] += n * T
    res["reward_mismatches"] += st["reward_mismatches"]
    res["done_mismatches"] += st["done_mismatches"]
    print(json.dumps(st), file=sys.stderr, flush=True)
res["max_obs_diff"] = max(s["max_obs_diff"] for s in res["bases"])
res["max_final_q_diff"] = max(s["max_final_q_diff"] for s in res["bases"])
res["min_flag_margin"] = min(min(s["flag_margins"].values()) for s in res["bases"])
json.dump(res, sys.stdout, indent=1)
print()
