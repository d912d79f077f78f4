"""Generate tests/golden/kuka_wrapper_reference.npz by running the REFERENCE's own
Kuka control / reward wrapper source
    /root/reference/environments/kuka_gym/kuka.py
    /root/reference/environments/kuka_gym/kuka_button_gym_env.py
against a SCRIPTED fake `pybullet` (the real pybullet==1.8.6 is not installed).

What this pins (everything the reference itself computes on the Kuka path):
  * RNG draw order and arithmetic of reset() / step(): the Cartesian IK target
    handed to p.calculateInverseKinematics after every applyAction (accumulate +
    clip, kuka.py:128-139) for seeded episodes and given discrete / continuous
    actions, including the 5 random init actions of reset();
  * the motor-command list of joint-space mode (kuka.py:160-163);
  * _reward / _termination bookkeeping (n_contacts, n_steps_outside, terminated,
    shaped rewards) on scripted gripper positions and contact flags.
  * EVERY p.setJointMotorControl2 call with all of its arguments (round 4: the `motorlog|...` arrays): the 14 position motors
    of Kuka.reset() (kuka.py:69-71), the seven arm motors + five gripper / tip motors of every applyAction (kuka.py:167-187)
    and the button motor of step2 (kuka_button_gym_env.py:347), as rows
        [body, joint, controlMode, targetPosition, targetVelocity, force, maxVelocity, positionGain, velocityGain]
    with NaN where the reference does not pass the argument (pybullet then uses its own default).
What it cannot pin: the physics inside pybullet (IK solution, dynamics, contacts).

Run in the build container only:  python tests/golden/make_kuka_wrapper_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _reference_stubs  # noqa: E402

PLANE, TABLE, BUTTON, KUKA = 0, 1, 2, 3


class Script(object):
    """Scripted physics outputs, indexed by the number of stepSimulation() calls."""

    def __init__(self):
        self.reset(None, None, None)

    def reset(self, gripper, contact_button, contact_table):
        self.gripper, self.contact_button, self.contact_table = gripper, contact_button, contact_table
        self.n_sim = 0
        self.ik_targets = []
        self.motor_targets = []
        self.button_motor_calls = 0
        self.motor_log = None            # a list: every setJointMotorControl2 call is appended as a 9-vector

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
        return uid                      # plane, table, button in call order

    def loadSDF(*a, **k):
        return [KUKA]

    def getNumJoints(uid):
        return 14

    def getJointInfo(uid, i):
        return (i, "joint{}".format(i).encode(), 0, 7 + i, 6 + i)

    def stepSimulation():
        SCRIPT.n_sim += 1

    def calculateInverseKinematics(uid, link, pos, orn=None, *a, **k):
        SCRIPT.ik_targets.append(np.array(pos, dtype=np.float64).copy())
        return [0.1 * (j + 1) for j in range(14)]

    def setJointMotorControl2(*a, **k):
        body = k.get("bodyUniqueId", a[0] if a else None)
        joint = k.get("jointIndex", a[1] if len(a) > 1 else None)
        if SCRIPT.motor_log is not None:
            mode = k.get("controlMode", a[2] if len(a) > 2 else np.nan)
            known = {"bodyUniqueId", "jointIndex", "controlMode", "targetPosition", "targetVelocity", "force", "maxVelocity",
                     "positionGain", "velocityGain"}
            assert set(k) <= known and len(a) <= 3, (a, k)      # nothing the log would drop
            SCRIPT.motor_log.append([float(body), float(joint), float(mode)] +
                                    [float(k.get(name, np.nan)) for name in ("targetPosition", "targetVelocity", "force", "maxVelocity",
                                                                             "positionGain", "velocityGain")])
        if body == BUTTON:
            SCRIPT.button_motor_calls += 1
        elif body == KUKA and joint is not None and joint <= 6 and "maxVelocity" in k:
            SCRIPT.motor_targets.append(float(k["targetPosition"]))

    def getLinkState(uid, link):
        if uid == BUTTON:
            return ((0.5, 0.0, -0.19), (0, 0, 0, 1))
        if SCRIPT.gripper is None:
            return ((0.5, 0.0, 0.4), (0, 0, 0, 1))
        return (tuple(SCRIPT.gripper[SCRIPT.idx()]), (0, 0, 0, 1))

    def getContactPoints(a, b, linkA=None):
        if SCRIPT.gripper is None:
            return []
        if a == BUTTON:
            return [1] if SCRIPT.contact_button[SCRIPT.idx()] else []
        return [1] if SCRIPT.contact_table[SCRIPT.idx()] else []

    p.resetSimulation, p.loadURDF, p.loadSDF = resetSimulation, loadURDF, loadSDF
    p.getNumJoints, p.getJointInfo, p.stepSimulation = getNumJoints, getJointInfo, stepSimulation
    p.calculateInverseKinematics, p.setJointMotorControl2 = calculateInverseKinematics, setJointMotorControl2
    p.getLinkState, p.getContactPoints = getLinkState, getContactPoints
    p.getQuaternionFromEuler = lambda e: (0.0, -1.0, 0.0, 0.0)
    p.getEulerFromQuaternion = lambda q: (0.0, 0.0, 0.0)
    return p


_reference_stubs.install(make_scripted_pybullet())

from environments.kuka_gym.kuka_button_gym_env import KukaButtonGymEnv  # noqa: E402
from environments.kuka_gym.kuka_moving_button_gym_env import KukaMovingButtonGymEnv  # noqa: E402

T = 1030          # crosses the 1001-step time limit


def action_case(seed, mode, random_target, force_down):
    """IK-target / motor-command trace with contact-free scripted physics."""
    kw = dict(srl_model="ground_truth", random_target=random_target, force_down=force_down)
    if mode == "discrete":
        kw["is_discrete"] = True
    elif mode == "continuous":
        kw["is_discrete"] = False
    else:
        kw.update(is_discrete=False, action_joints=True)
    env = KukaButtonGymEnv(**kw)
    env.seed(seed)
    arng = np.random.RandomState(777 + seed)
    SCRIPT.reset(None, None, None)
    env.reset()
    n_reset_sim = SCRIPT.n_sim
    out = {"reset_ik": np.array(SCRIPT.ik_targets[-5:]) if mode != "joints" else np.zeros((0, 3)),
           "reset_motor": np.array(SCRIPT.motor_targets[-35:]).reshape(5, 7) if mode == "joints" else np.zeros((0, 7)),
           "n_reset_sim": n_reset_sim, "button_pos": np.array(env.button_pos)}
    if mode == "discrete":
        actions = arng.randint(-1, 6, size=T)            # -1 stands for None
    elif mode == "continuous":
        # float64 copies of float32 values: `action[0] * dv` on a float32 *scalar* promotes differently in
        # numpy 1.14 (reference pin, -> float64) and numpy 2.x (-> float32); float64 inputs are unambiguous
        actions = arng.uniform(-1, 1, (T, 3)).astype(np.float32).astype(np.float64)
    else:
        actions = arng.uniform(-1, 1, (T, 7)).astype(np.float32)
    ik, motor, rewards, dones = [], [], [], []
    for t in range(T):
        SCRIPT.ik_targets, SCRIPT.motor_targets = [], []
        a = actions[t]
        if mode == "discrete":
            a = None if a < 0 else int(a)
        _, r, d, _ = env.step(a)
        ik.append(SCRIPT.ik_targets[-1] if SCRIPT.ik_targets else np.full(3, np.nan))
        motor.append(SCRIPT.motor_targets[-7:] if mode == "joints" else [np.nan] * 7)
        rewards.append(float(r))
        dones.append(bool(d))
        if d:
            break
    out.update(actions=np.asarray(actions), ik=np.array(ik), motor=np.array(motor), rewards=np.array(rewards),
               dones=np.array(dones), n_steps=len(dones))
    return out


def reward_case(seed, shape_reward, is_discrete, max_distance):
    """_reward/_termination bookkeeping on scripted gripper positions and contact flags."""
    env = KukaButtonGymEnv(srl_model="ground_truth", shape_reward=shape_reward, is_discrete=is_discrete,
                           max_distance=max_distance)
    env.seed(seed)
    srng = np.random.RandomState(4242 + seed)
    n = 1200
    # gripper wanders around the button target; a few button contacts, rare table contacts late
    grip = np.array([0.5, 0.0, 0.09]) + srng.normal(0, max_distance * 0.6, size=(n + 600, 3))
    cb = srng.rand(n + 600) < 0.004
    ct = (srng.rand(n + 600) < 0.002) & (np.arange(n + 600) > 900)
    SCRIPT.reset(None, None, None)
    env.reset()
    base = SCRIPT.n_sim
    SCRIPT.gripper, SCRIPT.contact_button, SCRIPT.contact_table = grip, cb, ct
    SCRIPT.n_sim = 0
    rec = {k: [] for k in ("reward", "done", "counter", "n_contacts", "n_outside", "terminated", "sim_idx")}
    for t in range(n):
        a = int(srng.randint(6)) if is_discrete else srng.uniform(-1, 1, 3).astype(np.float32)
        _, r, d, _ = env.step(a)
        rec["reward"].append(float(r)); rec["done"].append(bool(d)); rec["counter"].append(env._env_step_counter)
        rec["n_contacts"].append(env.n_contacts); rec["n_outside"].append(env.n_steps_outside)
        rec["terminated"].append(bool(env.terminated)); rec["sim_idx"].append(SCRIPT.idx())
        if d and t > 1100:
            break
    out = {k: np.array(v) for k, v in rec.items()}
    out.update(gripper=grip, contact_button=cb, contact_table=ct, button_pos=np.array(env.button_pos),
               n_reset_sim=base)
    return out


def moving_case(seed, shape_reward, random_target):
    """KukaMovingButtonGymEnv: direction draw, moving target, IK targets, reward branch, 1500-step limit."""
    env = KukaMovingButtonGymEnv(srl_model="ground_truth", shape_reward=shape_reward, random_target=random_target)
    env.seed(seed)
    srng = np.random.RandomState(99 + seed)
    n = 1560
    SCRIPT.reset(None, None, None)
    env.reset()
    speed0 = env.button_speed
    reset_ik = np.array(SCRIPT.ik_targets[-5:])
    grip = np.array([0.5, 0.0, 0.09]) + srng.normal(0, 0.3, size=(n + 10, 3))
    cb = srng.rand(n + 10) < 0.002
    ct = np.zeros(n + 10, bool)
    SCRIPT.gripper, SCRIPT.contact_button, SCRIPT.contact_table = grip, cb, ct
    SCRIPT.n_sim = 0
    actions = srng.randint(0, 4, size=n)                  # x / y moves only
    rec = {k: [] for k in ("reward", "done", "button_y", "ik", "sim_idx", "n_contacts")}
    for t in range(n):
        SCRIPT.ik_targets = []
        _, r, d, _ = env.step(int(actions[t]))
        rec["reward"].append(float(r)); rec["done"].append(bool(d)); rec["button_y"].append(float(env.button_pos[1]))
        rec["ik"].append(SCRIPT.ik_targets[-1] if SCRIPT.ik_targets else np.full(3, np.nan))
        rec["sim_idx"].append(SCRIPT.idx()); rec["n_contacts"].append(env.n_contacts)
        if d:
            break
    out = {k: np.array(v) for k, v in rec.items()}
    out.update(speed0=speed0, reset_ik=reset_ik, actions=actions, gripper=grip, contact_button=cb,
               button_pos0=np.array([env.button_pos[0], 0.0, env.button_pos[2]]), n_steps=len(rec["done"]))
    return out


def motorlog_case(mode):
    """Every setJointMotorControl2 call of one reset() and of two step() calls, all arguments (module docstring)."""
    kw = dict(srl_model="ground_truth")
    if mode == "discrete":
        kw["is_discrete"] = True
    elif mode == "continuous":
        kw["is_discrete"] = False
    else:
        kw.update(is_discrete=False, action_joints=True)
    env = KukaButtonGymEnv(**kw)
    env.seed(0)
    SCRIPT.reset(None, None, None)
    SCRIPT.motor_log = []
    env.reset()
    log_reset = np.array(SCRIPT.motor_log)
    n_reset_sim = SCRIPT.n_sim
    out = {"reset": log_reset, "n_reset_sim": n_reset_sim}
    for t in range(2):
        SCRIPT.motor_log = []
        if mode == "discrete":
            env.step(t + 2)
        elif mode == "continuous":
            env.step(np.array([0.25, -0.5, 0.75], dtype=np.float32).astype(np.float64))
        else:
            env.step(np.linspace(-1, 1, 7).astype(np.float32))
        out["step{}".format(t)] = np.array(SCRIPT.motor_log)
    SCRIPT.motor_log = None
    return out


def main():
    out = {}
    for mode in ("discrete", "continuous", "joints"):
        for k, v in motorlog_case(mode).items():
            out["motorlog|{}|{}".format(mode, k)] = v
    for seed in (0, 1, 2, 3):
        for shape_reward in (False, True):
            tag = "mov|s{}|sr{}|rt{}".format(seed, int(shape_reward), seed % 2)
            for k, v in moving_case(seed, shape_reward, bool(seed % 2)).items():
                out[tag + "|" + k] = v
    for seed in (0, 1, 2):
        for mode in ("discrete", "continuous", "joints"):
            for random_target in (False, True):
                for force_down in (True, False):
                    if mode == "joints" and not force_down:
                        continue
                    tag = "act|{}|s{}|rt{}|fd{}".format(mode, seed, int(random_target), int(force_down))
                    for k, v in action_case(seed, mode, random_target, force_down).items():
                        out[tag + "|" + k] = v
    for seed in (0, 1):
        for shape_reward in (False, True):
            for is_discrete in (True, False):
                for max_distance in (0.8, 0.28):
                    tag = "rew|s{}|sr{}|d{}|m{}".format(seed, int(shape_reward), int(is_discrete), max_distance)
                    for k, v in reward_case(seed, shape_reward, is_discrete, max_distance).items():
                        out[tag + "|" + k] = v
    path = os.path.join(HERE, "kuka_wrapper_reference.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, len(out), "arrays", os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
