#!/usr/bin/env python
"""How close the arm links WITHOUT collision proxies (iiwa links 0..4) come to the table and to the button — round-4 verdict,
"what's missing" #2: `kuka_button_gym_env.py:433-437` asks Bullet for contacts of ANY Kuka link, this repo's full model carries 16
spheres on links 5..11 only.  If the axis of every lower link stays further from the table top and from the button than the link's
hull is thick, those links can never contribute a contact and leaving them out changes no reward / termination flag.

CPU only (drives the oracle; TEST INFRASTRUCTURE):   python profiles/probes/lower_links_clearance.py > profiles/r05_lower_links_clearance.json
  * trajectories: the oracle's full model, default KukaButton configuration and random_target (the larger workspace box, kuka.py:46-53):
    random agent (256 envs x 2048 steps with auto-reset) and every scripted saturating policy of tests/kuka_scripts.py (discrete and
    continuous corners, joint-space +-1);
  * geometry: forward kinematics from the 510-double model table (tests/kuka_numpy_tree_ref.py's frames); the axis of link i is the
    segment from joint frame i to joint frame i + 1 (i = 0..4; link 0's segment starts at the base frame, kuka.py:63);
  * reported per link: the smallest height of its axis segment above the table top (model table `table_top_z`) and the smallest
    distance of the segment (9 sample points) to the button's BODY — a vertical cylinder of radius 0.10 m (the base, oracle/kuka_model.h
    KM_BASE_RADIUS; the cap is 0.09) from the table top up to 0.045 m above the button link's frame (cap height 0.03 + glider origin
    0.005 + travel 0.01), the frame being `button_pos` - (0, 0, 0.28) (kuka_button_gym_env.py:273-274) — over all sampled env-steps,
    next to the same figures for links 5 and 6 (which DO carry spheres).
The iiwa's links are ~0.06-0.09 m thick around their axes (recalled; the meshes are absent here)."""
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(REPO, "robotics-rl-srl_amd"), REPO, os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
import kuka_numpy_tree_ref as ref  # noqa: E402
import kuka_scripts  # noqa: E402
from oracle import kuka_clib  # noqa: E402
from srlhip import kuka_model  # noqa: E402
from srlhip import _lib  # noqa: E402  (host-side accessor of the baked model table; no GPU needed)

table = np.asarray(_lib.kuka_tree_default_model(), dtype=np.float64)
model = kuka_model.tree_to_dict(table)
J = ref.unpack(table)
TABLE_Z = float(model["table_top_z"])


def frames(q7):
    """joint-frame origins of the arm [N][8][3]: base, joints 0..6 (vectorised over N configurations; gripper joints do not move them)."""
    N = q7.shape[0]
    R = np.broadcast_to(np.eye(3), (N, 3, 3)).copy()
    p = np.broadcast_to(ref.BASE, (N, 3)).copy()
    out = [p.copy()]
    for i in range(7):
        j = J[i]
        assert j["parent"] == i - 1
        p = p + R @ j["xyz"]
        a = j["axis"]
        K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
        rot = np.eye(3)[None] + np.sin(q7[:, i])[:, None, None] * K[None] + (1 - np.cos(q7[:, i]))[:, None, None] * (K @ K)[None]
        R = R @ j["Rj"] @ rot
        out.append(p.copy())
    return np.stack(out, axis=1)


BUTTON_R, BUTTON_TOP = 0.10, 0.045 - 0.28        # body radius; top of the cap relative to button_pos


def seg_button_dist(a, b, button_pos):
    best = np.full(a.shape[0], 1e9)
    for t in np.linspace(0.0, 1.0, 9):
        x = a + t * (b - a)
        dxy = np.maximum(np.linalg.norm(x[:, :2] - button_pos[:, :2], axis=-1) - BUTTON_R, 0.0)
        dz = np.maximum(x[:, 2] - (button_pos[:, 2] + BUTTON_TOP), 0.0)
        best = np.minimum(best, np.hypot(dxy, dz))
    return best


def account(res, name, out):
    live = np.ones(out["done"].shape, bool)
    q = out["q"].reshape(-1, 7)
    button = (out["gripper"].astype(np.float64) - out["obs"][:, :, :3].astype(np.float64)).reshape(-1, 3)   # obs = gripper - button_pos
    P = frames(q)
    st = {"env_steps": int(live.sum())}
    for i in range(7):
        a, b = P[:, i], P[:, i + 1]                   # link i: frame i -> frame i + 1  (P[0] is the base, P[k + 1] joint k)
        st["link%d" % i] = {"min_height_above_table_top": float(np.minimum(a[:, 2], b[:, 2]).min() - TABLE_Z),
                            "min_dist_to_button_body": float(seg_button_dist(a, b, button).min())}
    res[name] = st


def main():
    kuka_clib.set_full(True)
    res = {"table_top_z": TABLE_Z, "base": ref.BASE.tolist(), "note": "link i = segment from joint frame i-1 (base for i = 0) to joint frame i; heights / distances in metres"}
    try:
        for rt in (False, True):
            tag = "random_target" if rt else "default"
            account(res, "random_agent/" + tag, kuka_clib.rollout(np.arange(256), 2048, random_target=rt, rng_mode=kuka_clib.RNG_MT19937))
            for nm, scripts, kw in (("discrete", kuka_scripts.discrete_scripts(), {}), ("continuous", kuka_scripts.continuous_scripts(), {"is_discrete": False}),
                                    ("joints", kuka_scripts.joint_scripts(), {"is_discrete": False, "action_joints": True})):
                names, seeds, actions = kuka_scripts.batch(scripts, (7, 8, 9, 10))
                account(res, "scripts_%s/%s" % (nm, tag), kuka_clib.rollout(seeds, kuka_scripts.T_SCRIPT, actions=actions, random_target=rt, **kw))
    finally:
        kuka_clib.set_full(False)
    worst = {}
    for k, st in res.items():
        if not isinstance(st, dict):
            continue
        for lk, v in st.items():
            if lk.startswith("link"):
                w = worst.setdefault(lk, {"min_height_above_table_top": 1e9, "min_dist_to_button_body": 1e9})
                for f in w:
                    w[f] = min(w[f], v[f])
    res["worst_over_everything"] = worst
    json.dump(res, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
