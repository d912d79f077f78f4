"""Generate tests/golden/kuka_2button_reference.npz by running the REFERENCE's own
    /root/reference/environments/kuka_gym/kuka_2button_gym_env.py  (+ kuka.py, kuka_button_gym_env.py)
against a SCRIPTED fake `pybullet` (the real pybullet==1.8.6 is not installed).

Pinned: reset()'s RNG draw order (the discarded draws of the first button, the second button's random
position), button_all_pos, the Cartesian IK targets of the 5 init actions and of every step (large workspace:
Kuka(small_constraints=False)), goal switching, n_contacts[2], the 4999-step safety counter, sparse and
shaped rewards, the 1500-step limit.  Not pinned: anything inside pybullet.

Run in the build container only:  python tests/golden/make_kuka_2button_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _reference_stubs  # noqa: E402

PLANE, TABLE, BUTTON1, BUTTON2, KUKA = 0, 1, 2, 3, 10


class Script(object):
    def __init__(self):
        self.reset(None, None, None)

    def reset(self, gripper, contact_buttons, contact_table):
        self.gripper, self.contact_buttons, self.contact_table = gripper, contact_buttons, contact_table
        self.n_sim = 0
        self.ik_targets = []
        self.ik_extra_args = None
        self.button_motor_calls = 0

    def idx(self):
        return max(self.n_sim - 1, 0)


SCRIPT = Script()


def make_scripted_pybullet():
    p = _reference_stubs.make_fake_pybullet()
    state = {"next_urdf": 0}

    def resetSimulation():
        state["next_urdf"] = 0

    def loadURDF(*a, **k):
        uid = state["next_urdf"]
        state["next_urdf"] += 1
        return uid                      # plane, table, button 1, button 2 in call order

    def stepSimulation():
        SCRIPT.n_sim += 1

    def calculateInverseKinematics(uid, link, pos, orn=None, *a, **k):
        SCRIPT.ik_targets.append(np.array(pos, dtype=np.float64).copy())
        SCRIPT.ik_extra_args = (len(a), sorted(k))       # the four null-space lists, no jointDamping
        return [0.1 * (j + 1) for j in range(14)]

    def setJointMotorControl2(*a, **k):
        body = k.get("bodyUniqueId", a[0] if a else None)
        if body in (BUTTON1, BUTTON2):
            SCRIPT.button_motor_calls += 1

    def getLinkState(uid, link):
        if SCRIPT.gripper is None:
            return ((0.5, 0.0, 0.4), (0, 0, 0, 1))
        return (tuple(SCRIPT.gripper[SCRIPT.idx()]), (0, 0, 0, 1))

    def getContactPoints(a, b, linkA=None):
        if SCRIPT.gripper is None:
            return []
        if a in (BUTTON1, BUTTON2):
            return [1] if SCRIPT.contact_buttons[a - BUTTON1][SCRIPT.idx()] else []
        return [1] if SCRIPT.contact_table[SCRIPT.idx()] else []

    p.resetSimulation, p.loadURDF, p.loadSDF = resetSimulation, loadURDF, lambda *a, **k: [KUKA]
    p.getNumJoints = lambda uid: 14
    p.getJointInfo = lambda uid, i: (i, "joint{}".format(i).encode(), 0, 7 + i, 6 + i)
    p.stepSimulation = stepSimulation
    p.calculateInverseKinematics, p.setJointMotorControl2 = calculateInverseKinematics, setJointMotorControl2
    p.getLinkState, p.getContactPoints = getLinkState, getContactPoints
    p.getQuaternionFromEuler = lambda e: (0.0, -1.0, 0.0, 0.0)
    p.getEulerFromQuaternion = lambda q: (0.0, 0.0, 0.0)
    return p


_reference_stubs.install(make_scripted_pybullet())

from environments.kuka_gym.kuka_2button_gym_env import Kuka2ButtonGymEnv  # noqa: E402

T = 1530          # crosses the 1501-step time limit


def action_case(seed, mode, random_target):
    kw = dict(srl_model="ground_truth", random_target=random_target, is_discrete=(mode == "discrete"))
    env = Kuka2ButtonGymEnv(**kw)
    env.seed(seed)
    arng = np.random.RandomState(555 + seed)
    SCRIPT.reset(None, None, None)
    obs0 = env.reset()
    out = {"reset_ik": np.array(SCRIPT.ik_targets[-5:]), "n_reset_sim": SCRIPT.n_sim,
           "button_all_pos": np.array(env.button_all_pos), "obs0": np.array(obs0),
           "ik_extra_positional_args": SCRIPT.ik_extra_args[0], "force_down": env._force_down,
           "max_distance": env._max_distance, "max_steps": env.max_steps}
    if mode == "discrete":
        actions = arng.randint(-1, 6, size=T)
    else:
        actions = arng.uniform(-1, 1, (T, 3)).astype(np.float32).astype(np.float64)
    ik, rewards, dones = [], [], []
    for t in range(T):
        SCRIPT.ik_targets = []
        a = actions[t]
        if mode == "discrete":
            a = None if a < 0 else int(a)
        _, r, d, _ = env.step(a)
        ik.append(SCRIPT.ik_targets[-1] if SCRIPT.ik_targets else np.full(3, np.nan))
        rewards.append(float(r)); dones.append(bool(d))
        if d:
            break
    out.update(actions=np.asarray(actions), ik=np.array(ik), rewards=np.array(rewards), dones=np.array(dones),
               n_steps=len(dones), button_motor_calls=SCRIPT.button_motor_calls)
    return out


def reward_case(seed, shape_reward, max_distance):
    env = Kuka2ButtonGymEnv(srl_model="ground_truth", shape_reward=shape_reward, max_distance=max_distance,
                            random_target=bool(seed % 2))
    env.seed(seed)
    srng = np.random.RandomState(31337 + seed)
    n = 1600
    SCRIPT.reset(None, None, None)
    env.reset()
    all_pos = np.array(env.button_all_pos)
    grip = np.array([0.5, 0.0, 0.09]) + srng.normal(0, max_distance * 0.55, size=(n, 3))
    # contacts: button 1 touched now and then early, button 2 later; occasional touches of the non-goal button too
    tt = np.arange(n)
    cb1 = srng.rand(n) < np.where(tt < 500, 0.02, 0.004)
    cb2 = srng.rand(n) < np.where(tt < 500, 0.004, 0.012 if seed < 2 else 0.003)
    ct = (srng.rand(n) < 0.002) & (tt > 1300)
    SCRIPT.gripper, SCRIPT.contact_buttons, SCRIPT.contact_table = grip, (cb1, cb2), ct
    SCRIPT.n_sim = 0
    rec = {k: [] for k in ("reward", "done", "counter", "n_contacts0", "n_contacts1", "n_outside", "terminated",
                           "goal_id", "sim_idx", "button_pos", "obs")}
    for t in range(n - 5):
        o, r, d, _ = env.step(int(srng.randint(6)))
        rec["reward"].append(float(r)); rec["done"].append(bool(d)); rec["counter"].append(env._env_step_counter)
        rec["n_contacts0"].append(env.n_contacts[0]); rec["n_contacts1"].append(env.n_contacts[1])
        rec["n_outside"].append(env.n_steps_outside); rec["terminated"].append(bool(env.terminated))
        rec["goal_id"].append(env.goal_id); rec["sim_idx"].append(SCRIPT.idx())
        rec["button_pos"].append(np.array(env.button_pos)); rec["obs"].append(np.array(o))
        if d:
            break
    out = {k: np.array(v) for k, v in rec.items()}
    out.update(gripper=grip, contact_b1=cb1, contact_b2=cb2, contact_table=ct, all_pos=all_pos)
    return out


def main():
    out = {}
    for seed in (0, 1, 2):
        for mode in ("discrete", "continuous"):
            for random_target in (False, True):
                tag = "act|{}|s{}|rt{}".format(mode, seed, int(random_target))
                for k, v in action_case(seed, mode, random_target).items():
                    out[tag + "|" + k] = v
    for seed in (0, 1, 2, 3):
        for shape_reward in (False, True):
            for max_distance in (2.0, 0.35):
                tag = "rew|s{}|sr{}|m{}".format(seed, int(shape_reward), max_distance)
                for k, v in reward_case(seed, shape_reward, max_distance).items():
                    out[tag + "|" + k] = v
    path = os.path.join(HERE, "kuka_2button_reference.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, len(out), "arrays", os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
