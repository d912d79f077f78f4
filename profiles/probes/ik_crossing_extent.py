#!/usr/bin/env python
"""How often the IK conditioning flag (SRLHIP_F_KUKA_IK_CROSSED) fires and what happens behind it — round-5 verdict item 1(c).
Runs on the GPU box:  python profiles/probes/ik_crossing_extent.py > gpurun_out/r05_ik_crossing_extent.json
  * random agent (device Philox), 4096 envs x 2048 steps with auto-reset — the headline configuration: flagged env-steps / total;
  * the scripted saturating policies of tests/kuka_scripts.py (seeds 7..10, default config, MT19937, one episode each): episodes that
    cross, env-steps before / behind the flag, max |dq| vs the oracle before / behind it, reward / done mismatches behind it.
TEST INFRASTRUCTURE (drives the oracle next to the product)."""
import contextlib
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(REPO, "robotics-rl-srl_amd"), REPO, os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: F401,E402  (before libsrlhip.so: __graft_entry__.build())
import kuka_scripts  # noqa: E402
import test_gpu_kuka_ik_crossing as tg  # noqa: E402
from oracle import kuka_clib  # noqa: E402
from srlhip import _lib  # noqa: E402

kuka_clib.set_full(True)
res = {}
n, T = 4096, 2048
cfg = _lib.default_config(_lib.ENV_KUKA_BUTTON)
cfg.num_envs, cfg.rng_mode, cfg.auto_reset = n, _lib.RNG_PHILOX, 1
h = _lib.Handle(cfg)
h.reset()
flagged = 0
for rep in range(4):                       # 4 rollouts = 3.4e7 env-steps
    h.rollout(T, want=())
fin = h.get_state(_lib.F_KUKA_IK_CROSSED)
res["random_agent"] = {"envs": n, "env_steps": 4 * n * T, "flagged_env_steps": int((fin >> 1).sum()), "envs_flagged_now": int((fin & 1).sum())}
h.close()
for name, scripts, kw in (("discrete", kuka_scripts.discrete_scripts(), {}), ("continuous", kuka_scripts.continuous_scripts(), {"is_discrete": False}),
                          ("joints", kuka_scripts.joint_scripts(), {"is_discrete": False, "action_joints": True})):
    with contextlib.redirect_stdout(sys.stderr):
        st, ora = tg.run(scripts, kuka_scripts.T_SCRIPT, **kw)
    names, ss, _ = kuka_scripts.batch(scripts, tg.SEEDS)
    st["episodes"] = len(names)
    st["crossing_scripts"] = sorted({nm.split("/")[0] for nm, c in zip(names, ora["ik_final"][:, 0]) if c})
    st["first_flag_step_min"] = int(kuka_scripts.first_index(ora["ik_crossed"] != 0).min())
    res["scripts_" + name] = st
json.dump(res, sys.stdout, indent=1)
print()
