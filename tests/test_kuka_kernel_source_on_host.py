"""CPU-side parity of the HIP Kuka stepper's arithmetic: the kernel's own source
(world-frame ABA, LDL^T IK, table-driven reset, FMA contraction) compiled for the
host vs the independent plain-C oracle (link-frame ABA, Gaussian elimination,
literal 505-step reset).  Tolerance 1e-4 on joints (north star); flags bit-exact."""
import numpy as np
import pytest

import hostcheck
from oracle import clib, kuka_clib

TOL = 1e-4


@pytest.fixture(scope="module", autouse=True)
def _build():
    clib.build()
    hostcheck.lib()


def compare(a, b):
    assert np.abs(a["q"] - b["q"]).max() <= TOL
    assert np.abs(a["gripper"] - b["gripper"]).max() <= TOL
    assert np.abs(a["obs"] - b["obs"]).max() <= TOL and np.abs(a["obs0"] - b["obs0"]).max() <= TOL
    assert np.array_equal(a["done"], b["done"])
    assert np.array_equal(a["ep_stats"][:, 1:], b["ep_stats"][:, 1:])          # episode length, count: exact
    # (shaped) return: a sum of up to 1501 distances, each within the 1e-4 position bar (typically 1e-8)
    assert (np.abs(a["ep_stats"][:, 0] - b["ep_stats"][:, 0]) <= 1e-6 * np.maximum(a["ep_stats"][:, 1], 1)).all()
    return np.abs(a["q"] - b["q"]).max()


def test_default_discrete_env_full_episodes():
    n, T = 12, 1100
    actions = np.random.RandomState(0).randint(6, size=(T, n)).astype(np.int32)
    a = kuka_clib.rollout(np.arange(n), T, actions=actions)
    b = hostcheck.rollout(np.arange(n), T, actions=actions)
    err = compare(a, b)
    assert np.array_equal(a["reward"], b["reward"])
    assert a["done"].sum() >= n          # every env crossed at least one reset
    assert err < 1e-8                     # what two independent f64 implementations actually achieve


@pytest.mark.parametrize("kw", [
    dict(random_target=True, shape_reward=True),
    dict(action_repeat=3, force_down=False, max_distance=0.28),
    dict(obs_mode=2, auto_reset=False),
    dict(rng_mode=kuka_clib.RNG_PHILOX, random_target=True),
])
def test_env_options(kw):
    n, T = 6, 500
    actions = np.random.RandomState(1).randint(-1, 6, size=(T, n)).astype(np.int32)     # includes None (-1)
    a = kuka_clib.rollout(100 + np.arange(n), T, actions=actions, **kw)
    b = hostcheck.rollout(100 + np.arange(n), T, actions=actions, **kw)
    compare(a, b)
    assert np.abs(a["reward64"] - b["reward64"]).max() <= TOL


@pytest.mark.parametrize("joints", [False, True])
def test_continuous_actions(joints):
    n, T, adim = 4, 300, 7 if joints else 3
    actions = np.random.RandomState(2).uniform(-1, 1, size=(T, n, adim)).astype(np.float32)
    kw = dict(is_discrete=False, action_joints=joints)
    a = kuka_clib.rollout(7 + np.arange(n), T, actions=actions, **kw)
    b = hostcheck.rollout(7 + np.arange(n), T, actions=actions, **kw)
    compare(a, b)


def test_device_sampled_actions_stream():
    a = kuka_clib.rollout(np.arange(5), 200, actions=None, rng_mode=kuka_clib.RNG_PHILOX)
    b = hostcheck.rollout(np.arange(5), 200, actions=None, rng_mode=kuka_clib.RNG_PHILOX)
    assert np.array_equal(a["actions"], b["actions"])
    compare(a, b)


def test_moving_button_variant():
    """KukaMovingButtonGymEnv semantics (kuka_moving_button_gym_env.py): direction draw first, button and target
    move 1 mm per step and bounce at |y| = 0.3, 1500-step limit, shaped reward branch for discrete actions."""
    n, T = 6, 1600
    actions = np.random.RandomState(3).randint(6, size=(T, n)).astype(np.int32)
    actions[:, 0] = -1                                                 # env 0 idles (None action): runs into the step limit
    try:
        kuka_clib.set_moving(True); hostcheck.set_moving(True)
        for kw in (dict(), dict(shape_reward=True, random_target=True)):
            a = kuka_clib.rollout(40 + np.arange(n), T, actions=actions, **kw)
            b = hostcheck.rollout(40 + np.arange(n), T, actions=actions, **kw)
            compare(a, b)
            assert np.abs(a["reward64"] - b["reward64"]).max() <= TOL
        assert a["ep_stats"][:, 1].max() == 1501                        # counter > 1500
    finally:
        kuka_clib.set_moving(False); hostcheck.set_moving(False)
