#!/usr/bin/env python
"""kuka_bullet_detail_sensitivity.py — how much do the RECALLED (not read: Bullet's source is absent here) details of
btMultiBodyConstraintSolver's sweep matter against the north-star bar (|dq| <= 1e-4, flags bit-exact)?  Oracle against oracle
(CPU, full model), on the workload shape of the PyBullet pin (seeds 0-2, RandomState(1234) discrete actions):
  bit 0  non-contact rows swept backwards on even iterations (`iteration & 1 ? j : size - 1 - j`)
  bit 1  non-contact rows in body-creation order (button before arm; per body joint-limit rows, then motors)
  bit 2  a second friction row per contact (SOLVER_USE_2_FRICTION_DIRECTIONS)
The projected Gauss-Seidel is NOT converged after 150 sweeps (the per-sweep |d dv| is still ~1e-3 in the worst step), so the
order of the rows is part of the result.  Output: one JSON object -> profiles/r03_kuka_bullet_detail_sensitivity.json.
Whoever runs tests/golden/make_kuka_pybullet_golden.py reads here which detail to settle first if the pin fails."""
import ctypes
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from oracle import kuka_clib  # noqa: E402


def main():
    kuka_clib.set_full(True)
    lib = kuka_clib._lib()
    lib.kuka_oracle_set_detail.argtypes = [ctypes.c_int]
    T, seeds = 1400, np.array([0, 1, 2])
    actions = np.random.RandomState(1234).randint(6, size=(T, len(seeds))).astype(np.int32)
    runs = {}
    for mask in (0, 1, 2, 3, 4):
        lib.kuka_oracle_set_detail(mask)
        runs[mask] = kuka_clib.rollout(seeds, T, actions=actions, aux=True)
    lib.kuka_oracle_set_detail(0)
    base = runs[0]
    out = {"workload": "KukaButtonGymEnv full model, seeds 0-2, %d steps of RandomState(1234) discrete actions, auto-reset" % T,
           "bar": 1e-4, "contact_env_steps": int((base["rows"][:, :, 0] > 0).sum()), "variants": {}}
    names = {1: "alternating sweep direction of the non-contact rows", 2: "non-contact rows in body-creation order",
             3: "both", 4: "two friction directions per contact"}
    for mask in (1, 2, 3, 4):
        r = runs[mask]
        same = (r["done"] == base["done"]) & (r["reward"] == base["reward"])
        first_flag_diff = [int(np.argmin(same[:, e])) if not same[:, e].all() else None for e in range(len(seeds))]
        dq = np.abs(r["q_all"] - base["q_all"]).max(axis=2)                      # [T][n]
        upto = [T if f is None else f for f in first_flag_diff]
        before = [float(dq[:upto[e], e].max()) if upto[e] else 0.0 for e in range(len(seeds))]
        free = base["rows"][:, :, 0] == 0
        first_contact = [int(np.argmax(~free[:, e])) if (~free[:, e]).any() else T for e in range(len(seeds))]
        out["variants"][names[mask]] = {
            "max_abs_dq_until_flags_differ": max(before),
            "max_abs_dq_before_first_contact": max(float(dq[:min(first_contact[e], upto[e]), e].max()) if min(first_contact[e], upto[e]) else 0.0
                                                   for e in range(len(seeds))),
            "first_step_with_different_reward_or_done": first_flag_diff,
            "reward_done_planes_identical": bool(same.all()),
        }
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
