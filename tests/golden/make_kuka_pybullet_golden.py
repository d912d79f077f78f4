#!/usr/bin/env python
"""make_kuka_pybullet_golden.py — the one command that pins the Kuka dynamics against PyBullet.

Run it where `import pybullet` works (pybullet==1.8.6 is what the reference pins, environment.yml:109) and a checkout of
the reference is available:

    python tests/golden/make_kuka_pybullet_golden.py [--reference /root/reference] [--out tests/golden/kuka_pybullet_reference.npz]

It drives the REFERENCE's own KukaButtonGymEnv (environments/kuka_gym/kuka_button_gym_env.py + kuka.py, unmodified; gym is
stubbed by tests/golden/_reference_stubs.py when it is not installed, with the restated gym==0.11.0 seeding) on the real
PyBullet, and records for seeds {0, 1, 2} x 2 episodes of RandomState(1234) discrete actions, per step:
joint positions / velocities of all 14 joints (7 arm + gripper), button glider position / velocity, gripper position (getArmPos),
button_pos, the two contact predicates of _reward (button link <-> arm, table <-> arm), reward, done, Kuka.end_effector_pos and
the observation.  It also extracts the FULL model table (`srlhip_kuka_tree_model`, 506 doubles: the 12-DoF arm + gripper tree)
from the loaded body through PyBullet's own introspection (srlhip.kuka_model.tree_from_pybullet: getJointInfo / getDynamicsInfo
/ getLinkState at q = 0) and the scene heights (table top / settled button base), plus the rounds 1-2 lumped table
(`srlhip_kuka_model`, 138 doubles) from the sdf file.

Round 4 — what decides the recalled SOLVER DETAILS (srlhip_kuka_tree_model.solver_detail / contact_erp / limit_erp / linear_slop;
tests/golden/fit_kuka_pin.py searches them on the oracle against this fixture) is recorded as well:
  * `settle_q14` / `settle_qd14` [seed][SETTLE_PROBE][14]: joint positions AND velocities after each of the first SETTLE_PROBE
    stepSimulation calls of every reset() — contact-free, RNG-free steps right after resetJointState: the cleanest trace of the
    sweep ORDER of the motor rows (an alternating sweep moves the arm joints by 2e-4 rad in the very first step);
  * `physics_engine_parameters`: p.getPhysicsEngineParameters() as JSON (erp, contactERP / erp2, linear slop, iterations ...);
  * per recorded step, for up to MAX_CONTACTS button <-> arm contact points: `contact_normal` [.][K][3], `contact_distance`,
    `contact_normal_force`, `contact_fric1` / `contact_fric2` (lateral friction impulses) and `contact_fric_dir1` / `contact_fric_dir2`
    — a non-zero second direction says SOLVER_USE_2_FRICTION_DIRECTIONS is in force (NaN where the installed pybullet's
    getContactPoints does not return the field).

tests/test_kuka_pybullet_pin.py then installs the full table in the oracle (oracle.kuka_clib.set_tree_model) and in the HIP
stepper (Handle.set_kuka_tree_model), replays the same seeds / actions and compares: joint positions within 1e-4, reward / done
flags bit-exact — the north-star bar.  Until this script has been run somewhere, that test SKIPS with "PARITY UNPINNED".

This container has no PyBullet: the script exits with status 2 and says so."""
import argparse
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
SETTLE_PROBE, MAX_CONTACTS = 10, 6


def engine_parameters(p):
    """p.getPhysicsEngineParameters() as a JSON string ("{}" when the call does not exist in this pybullet)."""
    import json
    try:
        d = p.getPhysicsEngineParameters()
    except Exception as exc:                     # noqa: BLE001  (older builds: no such call)
        return json.dumps({"error": str(exc)})
    return json.dumps({str(k): (v if isinstance(v, (int, float, str)) else str(v)) for k, v in dict(d).items()}, sort_keys=True)


def contact_record(p, body_a, body_b, link_a=None):
    """Up to MAX_CONTACTS contact points as fixed-size arrays: normal[K][3], distance[K], normal_force[K], fric1[K], dir1[K][3],
    fric2[K], dir2[K][3]; NaN-padded.  getContactPoints tuples: [7] contactNormalOnB, [8] contactDistance, [9] normalForce and,
    where the build returns them, [10] lateralFriction1, [11] lateralFrictionDir1, [12] lateralFriction2, [13] lateralFrictionDir2."""
    pts = p.getContactPoints(body_a, body_b, link_a) if link_a is not None else p.getContactPoints(body_a, body_b)
    K = MAX_CONTACTS
    rec = {"normal": np.full((K, 3), np.nan), "distance": np.full(K, np.nan), "normal_force": np.full(K, np.nan),
           "fric1": np.full(K, np.nan), "fric_dir1": np.full((K, 3), np.nan), "fric2": np.full(K, np.nan), "fric_dir2": np.full((K, 3), np.nan)}
    for i, c in enumerate(list(pts)[:K]):
        rec["normal"][i] = c[7]; rec["distance"][i] = c[8]; rec["normal_force"][i] = c[9]
        if len(c) > 13:
            rec["fric1"][i] = c[10]; rec["fric_dir1"][i] = c[11]; rec["fric2"][i] = c[12]; rec["fric_dir2"][i] = c[13]
    return rec, len(pts)


class SettleProbe(object):
    """Wraps p.stepSimulation for the duration of a reset(): joint states of `uid_of()` after each of the first SETTLE_PROBE calls."""

    def __init__(self, p, uid_of):
        self.p, self.uid_of, self.real = p, uid_of, p.stepSimulation
        self.q, self.qd = [], []

    def __enter__(self):
        def step(*a, **k):
            out = self.real(*a, **k)
            if len(self.q) < SETTLE_PROBE:
                js = [self.p.getJointState(self.uid_of(), j) for j in range(14)]
                self.q.append([s[0] for s in js]); self.qd.append([s[1] for s in js])
            return out
        self.p.stepSimulation = step
        return self

    def __exit__(self, *exc):
        self.p.stepSimulation = self.real
        return False


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(HERE, "kuka_pybullet_reference.npz"))
    ap.add_argument("--episodes", type=int, default=2)
    args = ap.parse_args()
    try:
        import pybullet as p
        import pybullet_data
    except ImportError as exc:
        print("PyBullet is not importable here ({}): the Kuka dynamics stay UNPINNED.  Run this script on a machine with "
              "pybullet==1.8.6 and commit the .npz it writes.".format(exc))
        return 2
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.join(REPO, "robotics-rl-srl_amd"))
    sys.path.insert(0, REPO)
    import _reference_stubs
    _reference_stubs.REFERENCE = args.reference
    try:
        import gym  # noqa: F401  the real one, if present
        sys.path.insert(0, args.reference)
    except ImportError:
        _reference_stubs.install(pybullet_module=p, real_pybullet_data=True)
    from environments.kuka_gym import kuka_button_gym_env as ref
    from srlhip import kuka_model

    # ---- model table from the files the reference loads (kuka.py:60, kuka_button_gym_env.py:221-223)
    sdf = os.path.join(pybullet_data.getDataPath(), "kuka_iiwa", "kuka_with_gripper2.sdf")
    model = kuka_model.from_sdf(sdf)

    tree = None
    records = {k: [] for k in ("seed", "episode", "action", "q", "qd", "q14", "qd14", "glider", "gripper", "button_pos", "contact_button",
                               "contact_table", "reward", "done", "ee_target", "obs", "obs0", "settle_q14", "settle_qd14", "n_contacts",
                               "contact_normal", "contact_distance", "contact_normal_force", "contact_fric1", "contact_fric_dir1",
                               "contact_fric2", "contact_fric_dir2")}
    params_json = None
    arng = np.random.RandomState(1234)
    for seed in (0, 1, 2):
        env = ref.KukaButtonGymEnv(renders=False, is_discrete=True, srl_model="ground_truth", record_data=False)
        env.seed(seed)
        for episode in range(args.episodes):
            with SettleProbe(p, lambda: env._kuka.kuka_uid) as probe:           # (env._kuka is created inside reset(), before its first step)
                obs = env.reset()
            records["settle_q14"].append(probe.q); records["settle_qd14"].append(probe.qd)
            if params_json is None:
                params_json = engine_parameters(p)
            records["obs0"].append(np.asarray(obs, dtype=np.float64))
            if seed == 0 and episode == 0:
                # scene heights as PyBullet settles them: top of the table, origin of the button's base link
                model["table_top_z"] = float(p.getAABB(env.table_uid)[1][2])
                model["button_base_z"] = float(p.getBasePositionAndOrientation(env.button_uid)[0][2])
                ls = p.getLinkState(env._kuka.kuka_uid, 6, computeForwardKinematics=True)      # link_7: world frame of the link
                R7 = np.array(p.getMatrixFromQuaternion(ls[5])).reshape(3, 3)
                p7 = np.array(ls[4])
                model["ee_point"] = R7.T @ (np.array(ls[0]) - p7)                                # its inertial frame = IK end effector
                model["gripper_point"] = R7.T @ (np.array(env.getArmPos()) - p7)
                tree = kuka_model.tree_from_pybullet(p, env._kuka.kuka_uid)
                tree["table_top_z"], tree["button_base_z"] = model["table_top_z"], model["button_base_z"]
            done = False
            while not done:
                a = int(arng.randint(6))
                obs, reward, done, _ = env.step(a)
                js = [p.getJointState(env._kuka.kuka_uid, j) for j in range(7)]
                gl = p.getJointState(env.button_uid, ref.BUTTON_GLIDER_IDX)
                records["seed"].append(seed); records["episode"].append(episode); records["action"].append(a)
                records["q"].append([s[0] for s in js]); records["qd"].append([s[1] for s in js])
                js14 = [p.getJointState(env._kuka.kuka_uid, j) for j in range(14)]
                records["q14"].append([s[0] for s in js14]); records["qd14"].append([s[1] for s in js14])
                records["glider"].append([gl[0], gl[1]])
                records["gripper"].append(list(env.getArmPos())); records["button_pos"].append(list(env.button_pos))
                records["contact_button"].append(int(len(p.getContactPoints(env.button_uid, env._kuka.kuka_uid, ref.BUTTON_LINK_IDX)) > 0))
                records["contact_table"].append(int(len(p.getContactPoints(env.table_uid, env._kuka.kuka_uid)) > 0))
                crec, ncon = contact_record(p, env.button_uid, env._kuka.kuka_uid)
                records["n_contacts"].append(ncon)
                for ck, cv in crec.items():
                    records["contact_" + ck].append(cv)
                records["reward"].append(float(reward)); records["done"].append(int(done))
                records["ee_target"].append(list(env._kuka.end_effector_pos)); records["obs"].append(np.asarray(obs, dtype=np.float64))
        env.close()
    out = {k: np.asarray(v) for k, v in records.items()}
    out["model_table"] = kuka_model.to_table(model)
    out["tree_model_table"] = kuka_model.tree_to_table(tree)
    out["physics_engine_parameters"] = np.array(params_json or "{}")
    out["pybullet_version"] = np.array(str(getattr(p, "getAPIVersion", lambda: "?")()))
    np.savez_compressed(args.out, **out)
    print("wrote {}: {} steps, model table from {}".format(args.out, len(out["action"]), sdf))
    return 0


if __name__ == "__main__":
    sys.exit(main())
