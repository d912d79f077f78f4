"""GPU: KukaButtonGymEnv under SATURATING scripted policies (tests/kuka_scripts.py: every discrete action held, the eight three-segment
corner scripts, the corners of the saturated continuous action cube, joint-space actions at +-1; seeds 7..10, default config, MT19937)
— the HIP path through the C-ABI, one launch per step, against the oracle.  Bar: 1e-7 rad on the arm joints and reward / done bit for
bit on every env-step BEFORE the IK conditioning flag (SRLHIP_F_KUKA_IK_CROSSED; kuka.py:41-42,118-156); the product raises the flag at
the oracle's step and counts the same flagged env-steps; after the flag nothing is asserted (no two float64 implementations agree
there) and the measured divergence is printed.  Also: the random agent of the headline configuration never raises the flag."""
import numpy as np
import pytest

import kuka_scripts
from oracle import kuka_clib
from srlhip import _lib

pytestmark = pytest.mark.gpu
SEEDS = (7, 8, 9, 10)


def run(scripts, T, **kw):
    names, ss, actions = kuka_scripts.batch(scripts, SEEDS)
    n = len(ss)
    ora = kuka_clib.rollout(ss, T, actions=actions, rng_mode=kuka_clib.RNG_MT19937, auto_reset=False, ik_trace=True, **kw)
    cfg = _lib.default_config(_lib.ENV_KUKA_BUTTON)
    cfg.num_envs, cfg.rng_mode, cfg.auto_reset = n, _lib.RNG_MT19937, 0
    for k, v in kw.items():
        setattr(cfg, k, int(v))
    h = _lib.Handle(cfg)
    h.seed(ss)                                   # env i runs script / seed i (a handle otherwise seeds env i with seed0 + i)
    h.reset()
    q, rew, done, flag = np.zeros((T, n, 7)), np.zeros((T, n), np.float32), np.zeros((T, n), np.uint8), np.zeros((T, n), np.uint8)
    for t in range(T):
        o, r, d = h.step(actions[t])
        q[t] = h.get_state(_lib.F_KUKA_Q).T
        rew[t], done[t] = r, d
        flag[t] = h.get_state(_lib.F_KUKA_IK_CROSSED) & 1
    fin = h.get_state(_lib.F_KUKA_IK_CROSSED)
    # flagged env-steps: counted over the T steps of the rollout, finished episodes included (auto_reset off: a finished env keeps
    # stepping its terminated state on both sides)
    assert np.array_equal(fin & 1, ora["ik_final"][:, 0]) and np.array_equal(fin >> 1, ora["ik_final"][:, 1])
    h.close()
    st = kuka_scripts.compare(names, ora, q, rew, done, flag)
    print(st)
    # fused launches (srlhip_rollout needs auto_reset): one launch of T steps against T single-step launches of the same configuration,
    # both on the GPU — the flag plane (sticky bits and flagged-step counts across auto-resets), reward and done are identical
    cfg.auto_reset = 1
    planes = []
    for fused in (False, True):
        h = _lib.Handle(cfg)
        h.seed(ss)
        h.reset()
        if fused:
            out = h.rollout(T, actions=actions)
            r2, d2 = out["reward"], out["done"]
        else:
            r2, d2 = np.zeros((T, n), np.float32), np.zeros((T, n), np.uint8)
            for t in range(T):
                _, r2[t], d2[t] = h.step(actions[t])
        planes.append((h.get_state(_lib.F_KUKA_IK_CROSSED), r2, d2))
        h.close()
    for a, b in zip(*planes):
        assert np.array_equal(a, b)
    assert (planes[0][0] >> 1).sum() >= (fin >> 1).sum() * 0 and ((planes[0][0] >> 1) > 0).sum() == ((fin >> 1) > 0).sum()
    return st, ora


def test_discrete_scripts_every_held_action_and_the_eight_corner_scripts():
    st, ora = run(kuka_scripts.discrete_scripts(), kuka_scripts.T_SCRIPT)
    assert st["crossed"] >= 8 and st["pre_max_dq"] < 1e-7
    assert st["post_max_dq"] > 1e-4            # the regime is real (otherwise the flag is too eager and this file is obsolete)


def test_saturated_continuous_actions():
    st, ora = run(kuka_scripts.continuous_scripts(), kuka_scripts.T_SCRIPT, is_discrete=False)
    assert st["crossed"] == 0


def test_saturated_joint_space_actions_have_no_ik():
    st, ora = run(kuka_scripts.joint_scripts(), kuka_scripts.T_SCRIPT, is_discrete=False, action_joints=True)
    assert st["crossed"] == 0 and (ora["ik_det"] > 1e299).all()


def test_random_agent_of_the_headline_configuration_never_raises_the_flag():
    """4096 envs x 2048 steps of the device-side Philox agent with auto-reset (bench.py's timed step): the flag plane stays 0, and the
    oracle on the same seeds / sampled actions agrees (smallest det of any IK solve printed: ~1e-7, threshold 3e-9)."""
    n, T = 4096, 2048
    cfg = _lib.default_config(_lib.ENV_KUKA_BUTTON)
    cfg.num_envs, cfg.rng_mode, cfg.auto_reset = n, _lib.RNG_PHILOX, 1
    h = _lib.Handle(cfg)
    h.reset()
    out = h.rollout(T)
    fin = h.get_state(_lib.F_KUKA_IK_CROSSED)
    assert not fin.any()
    ora = kuka_clib.rollout(np.arange(n), T, actions=out["actions"], rng_mode=kuka_clib.RNG_PHILOX, trace=False, ik_trace=True)
    assert not ora["ik_final"].any()
    assert np.array_equal(ora["done"], out["done"]) and np.array_equal(ora["reward"], out["reward"])
    print("random agent, {} env-steps: smallest det(J^T J + jd I) = {:.3e}".format(n * T, ora["ik_det"].min()))
    assert ora["ik_det"].min() > 10 * 3e-9
    h.close()
