"""CPU-side parity of the HIP Kuka steppers' arithmetic: the kernels' own source compiled for the host vs the
independent plain-C oracle (link-frame ABA, Gaussian elimination, literal 505-step reset).  Two kernels:
  lane   lane-per-env (kuka_core.hpp: world-frame ABA, LDL^T IK, table-driven reset, FMA contraction)
  group  16-lane group per env (kuka_group.hpp: prefix-sum RNEA / CRBA, Gauss-Jordan, one solver row per lane); its
         lanes run as lockstep fibers, every DPP primitive is an exchange between them.
Tolerance 1e-4 on joints (north star); flags bit-exact."""
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


STEPPERS = {"lane": hostcheck.rollout, "group": hostcheck.group_rollout}
both = pytest.mark.parametrize("stepper", ["lane", "group"])


@both
def test_default_discrete_env_full_episodes(stepper):
    n, T = 12, 1100
    actions = np.random.RandomState(0).randint(6, size=(T, n)).astype(np.int32)
    a = kuka_clib.rollout(np.arange(n), T, actions=actions)
    b = STEPPERS[stepper](np.arange(n), T, actions=actions)
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
@both
def test_env_options(kw, stepper):
    n, T = 6, 500
    actions = np.random.RandomState(1).randint(-1, 6, size=(T, n)).astype(np.int32)     # includes None (-1)
    a = kuka_clib.rollout(100 + np.arange(n), T, actions=actions, **kw)
    b = STEPPERS[stepper](100 + np.arange(n), T, actions=actions, **kw)
    compare(a, b)
    assert np.abs(a["reward64"] - b["reward64"]).max() <= TOL


@both
@pytest.mark.parametrize("joints", [False, True])
def test_continuous_actions(joints, stepper):
    n, T, adim = 4, 300, 7 if joints else 3
    actions = np.random.RandomState(2).uniform(-1, 1, size=(T, n, adim)).astype(np.float32)
    kw = dict(is_discrete=False, action_joints=joints)
    a = kuka_clib.rollout(7 + np.arange(n), T, actions=actions, **kw)
    b = STEPPERS[stepper](7 + np.arange(n), T, actions=actions, **kw)
    compare(a, b)


@both
def test_device_sampled_actions_stream(stepper):
    a = kuka_clib.rollout(np.arange(5), 200, actions=None, rng_mode=kuka_clib.RNG_PHILOX)
    b = STEPPERS[stepper](np.arange(5), 200, actions=None, rng_mode=kuka_clib.RNG_PHILOX)
    assert np.array_equal(a["actions"], b["actions"])
    compare(a, b)


@both
def test_moving_button_variant(stepper):
    """KukaMovingButtonGymEnv semantics (kuka_moving_button_gym_env.py): direction draw first, button and target
    move 1 mm per step and bounce at |y| = 0.3, 1500-step limit, shaped reward branch for discrete actions."""
    n, T = 6, 1600
    actions = np.random.RandomState(3).randint(6, size=(T, n)).astype(np.int32)
    actions[:, 0] = -1                                                 # env 0 idles (None action): runs into the step limit
    try:
        kuka_clib.set_moving(True); hostcheck.set_moving(True)
        for kw in (dict(), dict(shape_reward=True, random_target=True)):
            a = kuka_clib.rollout(40 + np.arange(n), T, actions=actions, **kw)
            b = STEPPERS[stepper](40 + np.arange(n), T, actions=actions, **kw)
            compare(a, b)
            assert np.abs(a["reward64"] - b["reward64"]).max() <= TOL
        assert a["ep_stats"][:, 1].max() == 1501                        # counter > 1500
    finally:
        kuka_clib.set_moving(False); hostcheck.set_moving(False)


def two_button_actions(n, T, seed=5):
    """random walks, plus scripted envs that press button 1 (y = +0.125), come back up and press button 2"""
    actions = np.random.RandomState(seed).randint(6, size=(T, n)).astype(np.int32)
    # the arm follows its IK target at <= 0.35 rad/s per joint (~1 mm per step): move, then wait with None actions
    script = [3] * 4 + [0] * 2 + [4] * 17 + [-1] * 420 + [5] * 10 + [-1] * 160 + [2] * 9 + [-1] * 260 + [4] * 10
    actions[:len(script), 0] = script
    actions[len(script):, 0] = -1
    actions[:, 1] = -1                                                  # idles into the 1500-step limit
    script2 = [2] * 4 + [4] * 50                                        # presses the wrong (second) button first
    actions[:len(script2), 2] = script2
    return actions


def test_two_button_variant():
    """Kuka2ButtonGymEnv (kuka_2button_gym_env.py): two button bodies, goal switching after 5 contacts with the first,
    episode ends after 5 contacts with the second, large workspace, default-damping IK, 1500-step limit."""
    n, T = 8, 1600
    actions = two_button_actions(n, T)
    try:
        kuka_clib.set_variant(2); hostcheck.set_variant(2)
        for kw in (dict(force_down=False, max_distance=2.0), dict(force_down=False, max_distance=2.0, shape_reward=True, random_target=True)):
            a = kuka_clib.rollout(60 + np.arange(n), T, actions=actions, **kw)
            b = hostcheck.rollout(60 + np.arange(n), T, actions=actions, **kw)
            compare(a, b)
            assert np.abs(a["reward64"] - b["reward64"]).max() <= TOL
            assert np.array_equal(a["final_state"][:, 26:28], b["final_state"][:, 26:28])      # goal_id, n_contacts[1]
            assert np.abs(a["final_state"][:, 24:26] - b["final_state"][:, 24:26]).max() <= TOL  # second glider
            assert np.array_equal(a["final_state"][:, 28:30], b["final_state"][:, 28:30])      # second button position
            if not kw.get("random_target"):
                first = np.argmax(a["done"][:, 0])
                assert a["done"][first, 0] and 900 < first < 1400 and a["reward"][first, 0] == 1.0   # both buttons pressed in order
                assert a["reward"][:first, 0].sum() == 4 and (a["reward"][:first, 0] != 0).sum() == 4   # sparse: last button only
        assert a["ep_stats"][:, 1].max() == 1501
    finally:
        kuka_clib.set_variant(0); hostcheck.set_variant(0)


@both
def test_rand_button_variant(stepper):
    """KukaRandButtonGymEnv: same stepping as KukaButtonGymEnv with the env's RNG stream shifted by the 20 distractor draws."""
    n, T = 6, 1100
    actions = np.random.RandomState(9).randint(6, size=(T, n)).astype(np.int32)
    base = kuka_clib.rollout(80 + np.arange(n), 40, actions=actions[:40], random_target=True)
    try:
        kuka_clib.set_variant(3); hostcheck.set_variant(3)
        a = kuka_clib.rollout(80 + np.arange(n), T, actions=actions, random_target=True)
        b = STEPPERS[stepper](80 + np.arange(n), T, actions=actions, random_target=True)
        compare(a, b)
        assert np.array_equal(a["reward"], b["reward"])
        assert np.abs(a["obs"][:40] - base["obs"]).max() > 1e-3       # a different episode than the base env's on the same seed
    finally:
        kuka_clib.set_variant(0); hostcheck.set_variant(0)
