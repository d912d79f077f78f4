"""Generate tests/golden/mobile_reference.npz by running the REFERENCE's own
MobileRobot env source (/root/reference/environments/mobile_robot/*.py) with
pybullet/gym stubbed (see _reference_stubs.py).

Run in the build container only:   python tests/golden/make_mobile_golden.py
The .npz is committed; tests never need /root/reference.

Protocol per case (mirrors SB VecEnv worker semantics):
  env.seed(seed); obs0 = env.reset(); then for every step: env.step(a);
  on done -> obs = env.reset() (auto-reset).  Actions are pre-drawn from
  RandomState(1234 + seed).  Recorded per step: obs (as returned, f64),
  reward, done, robot_pos (f64, after the step, before auto-reset),
  target (getTargetPos after the step), and the first obs of each episode.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _reference_stubs  # noqa: E402

_reference_stubs.install()

from environments.mobile_robot.mobile_robot_env import MobileRobotGymEnv  # noqa: E402
from environments.mobile_robot.mobile_robot_1D_env import MobileRobot1DGymEnv  # noqa: E402
from environments.mobile_robot.mobile_robot_2target_env import MobileRobot2TargetGymEnv  # noqa: E402
from environments.mobile_robot.mobile_robot_line_target_env import MobileRobotLineTargetGymEnv  # noqa: E402

KINDS = {
    "mobile": (MobileRobotGymEnv, 4),
    "mobile1d": (MobileRobot1DGymEnv, 2),
    "mobile2t": (MobileRobot2TargetGymEnv, 4),
    "mobileline": (MobileRobotLineTargetGymEnv, 4),
}
N_STEPS = 2 * 251 + 7          # crosses two auto-resets
SEEDS = (0, 1, 2)


def run_case(cls, n_act, seed, random_target, shape_reward, continuous):
    env = cls(srl_model="ground_truth", is_discrete=not continuous,
              random_target=random_target, shape_reward=shape_reward)
    env.seed(seed)
    arng = np.random.RandomState(1234 + seed)
    if continuous:
        actions = arng.uniform(-1.5, 1.5, (N_STEPS, 2)).astype(np.float32)
    else:
        actions = arng.randint(n_act, size=N_STEPS).astype(np.int32)
    obs0 = np.asarray(env.reset(), dtype=np.float64)
    rec = {k: [] for k in ("obs", "reward", "done", "pos", "target", "reset_obs")}
    for t in range(N_STEPS):
        a = actions[t] if continuous else int(actions[t])
        obs, reward, done, _ = env.step(a)
        rec["obs"].append(np.asarray(obs, dtype=np.float64))
        rec["reward"].append(float(reward))
        rec["done"].append(bool(done))
        rec["pos"].append(np.array(env.robot_pos, dtype=np.float64))
        rec["target"].append(np.array(env.getTargetPos(), dtype=np.float64))
        if done:
            rec["reset_obs"].append(np.asarray(env.reset(), dtype=np.float64))
    out = {"actions": actions, "obs0": obs0}
    for k, v in rec.items():
        out[k] = np.array(v)
    return out


def main():
    out = {}
    for kind, (cls, n_act) in KINDS.items():
        for seed in SEEDS:
            for random_target in (False, True):
                for shape_reward in (False, True):
                    modes = [False]
                    if kind in ("mobile", "mobileline") and not shape_reward:
                        modes.append(True)      # continuous actions (a3)
                    for continuous in modes:
                        tag = "{}|s{}|rt{}|sr{}|c{}".format(
                            kind, seed, int(random_target), int(shape_reward), int(continuous))
                        case = run_case(cls, n_act, seed, random_target, shape_reward, continuous)
                        for k, v in case.items():
                            out[tag + "|" + k] = v
    path = os.path.join(HERE, "mobile_reference.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, len(out), "arrays", os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
