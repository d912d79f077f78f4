"""ctypes loader for oracle/_build/liboracle.so (plain-C oracle).
TEST INFRASTRUCTURE ONLY — see oracle/__init__.py."""
import ctypes
import os
import subprocess

import numpy as np

from . import gym_seeding

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_build", "liboracle.so")
_lib = None

RNG_PHILOX, RNG_MT19937 = 1, 2


def build(force=False):
    """gcc the C restatement (seconds)."""
    if force or not os.path.exists(LIB_PATH) or any(
            os.path.getmtime(os.path.join(HERE, f)) > os.path.getmtime(LIB_PATH)
            for f in os.listdir(HERE) if f.endswith((".c", ".h"))):
        subprocess.check_call(["make", "-s", "-C", HERE])
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = ctypes.CDLL(LIB_PATH)
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def mt_keys(seeds):
    """gym hash digits for each integer seed -> (keys [n][2] u32, key_len [n] i32)."""
    keys = np.zeros((len(seeds), 2), dtype=np.uint32)
    lens = np.zeros(len(seeds), dtype=np.int32)
    for i, s in enumerate(seeds):
        d = gym_seeding.hash_seed_digits(int(s))
        lens[i] = len(d)
        keys[i, :len(d)] = d
    return keys, lens


def mobile_rollout(kind, seeds, T, actions=None, is_discrete=True, random_target=False, shape_reward=False,
                   rng_mode=RNG_MT19937):
    """-> dict(obs0, obs, reward, reward64, done, actions, final_state, ep_stats)."""
    seeds = np.ascontiguousarray(seeds, dtype=np.int64)
    n = len(seeds)
    od = 1 if kind == 1 else 2
    keys, lens = mt_keys(seeds) if rng_mode == RNG_MT19937 else (np.zeros((n, 2), np.uint32), np.zeros(n, np.int32))
    out = {
        "obs0": np.zeros((n, od), np.float32), "obs": np.zeros((T, n, od), np.float32),
        "reward": np.zeros((T, n), np.float32), "reward64": np.zeros((T, n), np.float64),
        "done": np.zeros((T, n), np.uint8), "final_state": np.zeros((n, 8), np.float64),
        "ep_stats": np.zeros((n, 3), np.float64),
    }
    act_out = None
    if actions is None:
        act_out = np.zeros((T, n), np.int32) if is_discrete else np.zeros((T, n, 2), np.float32)
    else:
        actions = np.ascontiguousarray(actions, dtype=np.int32 if is_discrete else np.float32)
        assert actions.shape[:2] == (T, n)
    rc = lib().mobile_oracle_rollout(
        int(kind), int(is_discrete), int(random_target), int(shape_reward), int(rng_mode), n, int(T),
        _p(seeds), _p(keys), _p(lens), _p(actions), _p(out["obs0"]), _p(out["obs"]), _p(out["reward"]),
        _p(out["reward64"]), _p(out["done"]), _p(act_out), _p(out["final_state"]), _p(out["ep_stats"]))
    assert rc == 0, rc
    out["actions"] = actions if actions is not None else act_out
    return out


def np_random_draws(key, n):
    key = np.ascontiguousarray(key, dtype=np.uint32)
    u, g, r3 = np.zeros(n), np.zeros(n), np.zeros(n, np.uint32)
    lib().oracle_np_random_draws(_p(key), len(key), n, _p(u), _p(g), _p(r3))
    return u, g, r3
