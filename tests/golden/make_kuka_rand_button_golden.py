"""Generate tests/golden/kuka_rand_button_reference.npz by running the REFERENCE's own
    /root/reference/environments/kuka_gym/kuka_rand_button_gym_env.py  (+ kuka.py, kuka_button_gym_env.py)
against a SCRIPTED fake `pybullet`.

Pinned: the 20 extra np_random draws of reset() (two per candidate distractor, drawn whether or not the object is
kept), the keep rule (outside the 0.2 m square around the button), their positions, the ball's start position, and
— through the shifted RNG stream — the init actions / step noise that follow (IK target traces).
Not pinned: the object TYPES and the push on the ball (drawn from the global, unseeded np.random in the reference)
and all rigid-body dynamics of those bodies.

Run in the build container only:  python tests/golden/make_kuka_rand_button_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _reference_stubs  # noqa: E402

KUKA = 100
LOADS = []
IK = []
FORCES = []


def make_scripted_pybullet():
    p = _reference_stubs.make_fake_pybullet()
    state = {"next": 0}

    def resetSimulation():
        state["next"] = 0
        del LOADS[:]

    def loadURDF(path, *a, **k):
        uid = state["next"]
        state["next"] += 1
        pos = a[0] if a and isinstance(a[0], (list, tuple)) else (a[:3] if len(a) >= 3 else None)
        LOADS.append((os.path.basename(str(path)), None if pos is None else tuple(float(v) for v in pos)))
        return uid

    def calculateInverseKinematics(uid, link, pos, orn=None, *a, **k):
        IK.append(np.array(pos, dtype=np.float64).copy())
        return [0.1 * (j + 1) for j in range(14)]

    def getLinkState(uid, link):
        if uid == KUKA:
            return ((0.5, 0.0, 0.4), (0, 0, 0, 1))
        return ((0.5, 0.0, -0.19), (0, 0, 0, 1))          # button cap

    p.resetSimulation, p.loadURDF, p.loadSDF = resetSimulation, loadURDF, lambda *a, **k: [KUKA]
    p.getNumJoints = lambda uid: 14
    p.getJointInfo = lambda uid, i: (i, "joint{}".format(i).encode(), 0, 7 + i, 6 + i)
    p.calculateInverseKinematics, p.getLinkState = calculateInverseKinematics, getLinkState
    p.getContactPoints = lambda *a, **k: []
    p.applyExternalForce = lambda uid, link, force, pos, frame: FORCES.append(np.array(force, dtype=np.float64))
    p.WORLD_FRAME = 1
    p.getQuaternionFromEuler = lambda e: (0.0, -1.0, 0.0, 0.0)
    p.getEulerFromQuaternion = lambda q: (0.0, 0.0, 0.0)
    return p


_reference_stubs.install(make_scripted_pybullet())

from environments.kuka_gym.kuka_rand_button_gym_env import KukaRandButtonGymEnv  # noqa: E402

T = 60


def case(seed, random_target):
    np.random.seed(1000 + seed)                    # the reference's global draws (types, push): arbitrary, not compared
    env = KukaRandButtonGymEnv(srl_model="ground_truth", random_target=random_target)
    env.seed(seed)
    del IK[:], FORCES[:]
    env.reset()
    objs = [(n, pos) for n, pos in LOADS if n in ("duck_vhacd.urdf", "lego.urdf", "cube_small.urdf")]
    ball = [pos for n, pos in LOADS if n == "sphere_small.urdf"]
    button = [pos for n, pos in LOADS if n == "simple_button.urdf"]
    out = {"n_objects": len(objs), "object_pos": np.array([pos for _, pos in objs]).reshape(-1, 3), "ball_pos": np.array(ball[0]),
           "button_xy": np.array(button[0][:2]), "reset_ik": np.array(IK[-5:]), "max_steps": env.max_steps}
    actions = np.random.RandomState(seed).randint(6, size=T)
    ik = []
    for t in range(T):
        del IK[:]
        env.step(int(actions[t]))
        ik.append(IK[-1].copy())
    out.update(actions=actions, ik=np.array(ik), n_pushes=len(FORCES), push=np.array(FORCES[0]) if FORCES else np.zeros(3))
    return out


def main():
    out = {}
    for seed in range(6):
        for rt in (False, True):
            for k, v in case(seed, rt).items():
                out["s{}|rt{}|{}".format(seed, int(rt), k)] = v
    path = os.path.join(HERE, "kuka_rand_button_reference.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, len(out), "arrays", os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
