"""Model-gap report: how far is the rounds 1-2 Kuka model (gripper lumped rigidly into link_7, six gripper
spheres, frictionless contacts) from the full model (12-DoF gripper tree with the reference's finger / tip motors, 16 contact
spheres on links 5..11, one friction row per contact) — oracle vs oracle, same seeds, same actions.

    python profiles/probes/kuka_model_gap.py > profiles/r03_kuka_model_gap.json

Seeds {0, 1, 2} (env i seeded seed0 + i like makeEnv, the reference's MT19937 streams), actions pre-drawn with
RandomState(1234), two episodes per env (auto-reset).  Episodes are compared step by step while their reward / done planes
agree; after the first flag difference the trajectories are no longer the same experiment."""
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from oracle import kuka_clib  # noqa: E402

T, SEEDS = 2100, np.arange(3)
actions = np.random.RandomState(1234).randint(6, size=(T, len(SEEDS))).astype(np.int32)


def run(full, friction=True):
    kuka_clib.set_full(full)
    if full and not friction:
        t = kuka_clib.get_tree_model()
        t[-1] = 0.0                                   # tree_model.friction
        kuka_clib.set_tree_model(t)
    out = kuka_clib.rollout(SEEDS, T, actions=actions, rng_mode=kuka_clib.RNG_MT19937, aux=True)
    kuka_clib.set_full(False)
    return out


def compare(a, b):
    res = []
    for i in range(len(SEEDS)):
        da, db = a["done"][:, i], b["done"][:, i]
        flag_diff = np.nonzero((a["reward"][:, i] != b["reward"][:, i]) | (da != db))[0]
        first = int(flag_diff[0]) if len(flag_diff) else None
        upto = first if first is not None else T
        dq = np.abs(a["q"][:upto, i] - b["q"][:upto, i])
        dg = np.abs(a["gripper"][:upto, i] - b["gripper"][:upto, i])
        ends_a, ends_b = np.nonzero(da)[0][:2].tolist(), np.nonzero(db)[0][:2].tolist()
        res.append({"seed": int(SEEDS[i]), "first_step_with_different_reward_or_done": first,
                    "steps_compared": int(upto),
                    "max_abs_dq_arm": float(dq.max()) if upto else None,
                    "max_abs_dq_arm_first_100_steps": float(dq[:100].max()) if upto else None,
                    "max_abs_d_getArmPos": float(dg.max()) if upto else None,
                    "episode_end_steps": {"a": ends_a, "b": ends_b},
                    "episode_returns": {"a": [float(a["reward"][:e + 1, i].sum()) for e in ends_a[:1]],
                                        "b": [float(b["reward"][:e + 1, i].sum()) for e in ends_b[:1]]},
                    "contact_normal_rows": {"a": int(a["rows"][:, i, 0].sum()), "b": int(b["rows"][:, i, 0].sum())},
                    "friction_rows": {"a": int(a["rows"][:, i, 1].sum()), "b": int(b["rows"][:, i, 1].sum())},
                    "steps_with_button_reward": {"a": int((a["reward"][:, i] == 1).sum()), "b": int((b["reward"][:, i] == 1).sum())}})
    return res


lumped, full, full_nofric = run(False), run(True), run(True, friction=False)
report = {
    "what": "oracle (oracle/kuka_oracle.c) vs oracle: lumped-gripper 7-DoF model of rounds 1-2 (a) against the full 12-DoF gripper tree (b)",
    "protocol": "KukaButtonGymEnv-v0 defaults, seeds 0..2 (MT19937 streams), actions RandomState(1234).randint(6), T = %d steps with auto-reset" % T,
    "lumped_vs_full": compare(lumped, full),
    "full_without_friction_vs_full": compare(full_nofric, full),
    "gripper_joint_range_in_full_model": {"min": full["q_all"][:, :, 7:].min(axis=(0, 1)).tolist(), "max": full["q_all"][:, :, 7:].max(axis=(0, 1)).tolist(),
                                          "dofs": "gripper_to_arm, left finger, left tip, right finger, right tip"},
}
worst = max(r["max_abs_dq_arm"] or 0.0 for r in report["lumped_vs_full"])
report["verdict"] = ("max |dq| on the arm joints = %.3g rad (bar: 1e-4) and the reward / done planes differ: the lumped model cannot meet the "
                     "north-star tolerance against a simulator that integrates the gripper -> the full model is implemented in the stepper" % worst)
print(json.dumps(report, indent=1))
