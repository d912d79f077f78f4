"""CPU oracle for the MobileRobot env family — TEST INFRASTRUCTURE ONLY.

Scalar float64 restatement of the reference's kinematic stepper:
  * /root/reference/environments/mobile_robot/mobile_robot_env.py
      constants :13-28, arena/margins :101-104, reset RNG order :159-181,
      step :235-280, _termination :336-343, _reward :345-363
  * mobile_robot_1D_env.py:58-74 (reset), :108-147 (step), :149-168 (reward)
  * mobile_robot_2target_env.py:35-67 (reset), :114-116 (target), :162-181 (reward)
  * mobile_robot_line_target_env.py:3-4 (constants), :35-40 (target), :56-64
    (reset), :108-125 (reward)
RNG: gym_seeding.np_random (srl_env.py:71-78) -> numpy RandomState.

Pinned by tests/test_oracle_golden.py against tests/golden/mobile_reference.npz
(vectors produced by the reference source itself).
"""
import ctypes
import ctypes.util
import math

import numpy as np

from . import gym_seeding

MOBILE, MOBILE_1D, MOBILE_2TARGET, MOBILE_LINE = 0, 1, 2, 3
KIND_NAMES = {"mobile": MOBILE, "mobile1d": MOBILE_1D, "mobile2t": MOBILE_2TARGET,
              "mobileline": MOBILE_LINE}

MAX_STEPS = 250                  # mobile_robot_env.py:14 (2Target's 1500 is dead)
DELTA_POS = 0.1                  # :22
NOISE_STD = 0.0                  # :24
ROBOT_WIDTH = 0.2                # :29
ROBOT_LENGTH = 0.325 * 2         # :30
MAX_X = MAX_Y = 4                # :101-102
COLLISION_MARGIN = 0.1           # :104
ROBOT_OFFSET = 0.2               # line_target:4


_libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
_libm.fma.restype = ctypes.c_double
_libm.fma.argtypes = [ctypes.c_double] * 3


def norm2(*v):
    """``np.linalg.norm(v, 2)`` of a short float64 vector as the reference
    evaluates it (mobile_robot_env.py:350): sqrt(dot(v, v)) where BLAS ddot
    accumulates with fused multiply-adds, acc = fma(v[i], v[i], acc) from 0.
    Verified bit-exact against numpy/OpenBLAS on 20 000 random vectors
    (tests/test_oracle_golden.py::test_norm2_matches_numpy)."""
    acc = 0.0
    for c in v:
        acc = _libm.fma(c, c, acc)
    return math.sqrt(acc)


def n_actions(kind):
    return 2 if kind == MOBILE_1D else 4


def obs_dim(kind):
    return 1 if kind == MOBILE_1D else 2


class MobileOracleEnv(object):
    """One env; float64 throughout, Python control flow as in the reference."""

    def __init__(self, kind=MOBILE, is_discrete=True, random_target=False, shape_reward=False):
        self.kind = kind
        self.is_discrete = is_discrete
        self.random_target = random_target
        self.shape_reward = shape_reward
        self.reward_threshold = 0.1 if kind == MOBILE_LINE else 0.4
        self.np_random = None
        self.seed(0)                                    # srl_env.py:31
        self.x = self.y = 0.0
        self.targets = [(0.0, 0.0)]
        self.current_target = 0
        self.counter = 0
        self.has_bumped = False

    def seed(self, seed):
        self.np_random, seed = gym_seeding.np_random(seed)
        return [seed]

    # ---- getters (mobile_robot_env.py:147-157 and variant overrides) -------
    def target_pos(self):
        tx, ty = self.targets[self.current_target]
        if self.kind == MOBILE_1D:
            return np.array([tx])
        if self.kind == MOBILE_LINE:
            return np.array([tx]) - ROBOT_OFFSET
        return np.array([tx, ty])

    def ground_truth(self):
        if self.kind == MOBILE_1D:
            return np.array([self.x])
        return np.array([self.x, self.y])

    def obs(self):
        # srl_env.py:39-42 with RELATIVE_POS=True; numpy broadcasting makes the
        # LineTarget obs [x - tx', y - tx'].
        return self.ground_truth() - self.target_pos()

    # ---- reset --------------------------------------------------------------
    def reset(self):
        rng = self.np_random
        self.current_target = 0
        self.x = MAX_X / 2 + rng.uniform(-MAX_X / 3, MAX_X / 3)
        if self.kind == MOBILE_1D:
            self.y = 0.0
        else:
            self.y = MAX_Y / 2 + rng.uniform(-MAX_Y / 3, MAX_Y / 3)
        margin = 0.1 * MAX_X
        if self.kind == MOBILE_1D:
            tx = 0.9 * MAX_X
            if self.random_target:
                tx = rng.uniform(0 + margin, MAX_X - margin)
            self.targets = [(tx, 0.0)]
        elif self.kind == MOBILE_LINE:
            tx = 0.9 * MAX_X
            if self.random_target:
                tx = rng.uniform(0 + margin, MAX_X - margin)
            self.targets = [(tx, float(MAX_X))]
        else:
            tx, ty = 0.9 * MAX_X, MAX_Y * 3 / 4
            if self.random_target:
                tx = rng.uniform(0 + margin, MAX_X - margin)
                ty = rng.uniform(0 + margin, MAX_Y - margin)
            self.targets = [(tx, ty)]
            if self.kind == MOBILE_2TARGET:
                tx, ty = 0.1 * MAX_X, MAX_Y * 3 / 4
                if self.random_target:
                    tx = rng.uniform(0 + margin, MAX_X - margin)
                    ty = rng.uniform(0 + margin, MAX_Y - margin)
                self.targets.append((tx, ty))
        self.counter = 0
        return self.obs()

    # ---- step ---------------------------------------------------------------
    def step(self, action):
        self.has_bumped = False
        dv = DELTA_POS + self.np_random.normal(0.0, scale=NOISE_STD)   # drawn even at scale 0
        if self.is_discrete:
            if self.kind == MOBILE_1D:
                dx, dy = [-dv, dv][action], 0.0
            else:
                dx = [-dv, dv, 0, 0][action]
                dy = [0, 0, -dv, dv][action]
        else:
            if self.kind in (MOBILE_1D, MOBILE_2TARGET):
                raise ValueError("Only discrete actions is supported")
            # float32 Box action * python float stays float32 (numpy scalar casting)
            a = np.maximum(np.minimum(np.asarray(action), 1), -1) * dv
            dx, dy = float(a[0]), float(a[1])
        px, py = self.x, self.y
        self.x = self.x + dx
        if self.kind != MOBILE_1D:
            self.y = self.y + dy
        checks = [(self.x, MAX_X, ROBOT_LENGTH)]
        if self.kind != MOBILE_1D:
            checks.append((self.y, MAX_Y, ROBOT_WIDTH))
        for value, limit, dim in checks:
            margin = COLLISION_MARGIN + dim / 2
            if value < margin or value > limit - margin:
                self.has_bumped = True
                self.x, self.y = px, py
                break
        self.counter += 1
        reward = self._reward()
        done = self.counter > MAX_STEPS
        return self.obs(), reward, done

    def _reward(self):
        tgt = self.target_pos()
        if self.kind == MOBILE_LINE:
            distance = abs(tgt[0] - self.x)
        elif self.kind == MOBILE_1D:
            distance = norm2(tgt[0] - self.x)
        else:
            distance = norm2(tgt[0] - self.x, tgt[1] - self.y)
        reward = 0
        if distance <= self.reward_threshold:
            reward = 1
            if self.kind == MOBILE_2TARGET and self.current_target < len(self.targets) - 1:
                self.current_target += 1
        if self.has_bumped:
            reward = -1
        if self.shape_reward:
            return -distance
        return reward


def rollout(kind, seed, actions, is_discrete=True, random_target=False, shape_reward=False):
    """Auto-resetting rollout of one env (VecEnv worker semantics).  Returns a
    dict with the same fields as tests/golden/make_mobile_golden.py."""
    env = MobileOracleEnv(kind, is_discrete, random_target, shape_reward)
    env.seed(seed)
    rec = {k: [] for k in ("obs", "reward", "done", "pos", "target", "reset_obs")}
    obs0 = env.reset()
    for a in actions:
        obs, reward, done = env.step(a if not is_discrete else int(a))
        rec["obs"].append(obs)
        rec["reward"].append(float(reward))
        rec["done"].append(done)
        rec["pos"].append([env.x, env.y, 0.0])
        rec["target"].append(env.target_pos())
        if done:
            rec["reset_obs"].append(env.reset())
    out = {k: np.array(v) for k, v in rec.items()}
    out["obs0"] = obs0
    return out
