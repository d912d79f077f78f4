"""GPU parity of the two-wavefronts-per-SIMD variant of the full-model Kuka kernel (csrc/kuka_tree_occ.hip): the same step
functions (csrc/kuka_tree.hpp, OCC = 1) with the contact-candidate list recomputed instead of kept in registers, the generic
path's work area shared by the four envs of a wavefront (they take turns) and the per-env state parked in LDS meanwhile.  The
library picks it for one-button envs with Cartesian actions from 65536 envs up; SRLHIP_KUKA_OCC=1 forces it on the small batches
the oracle finishes in seconds, =0 forces the one-wavefront kernel.  Bar: the north star's (1e-4 on joints, flags bit for bit)."""
import numpy as np
import pytest

from oracle import kuka_clib
from srlhip import _lib

from test_gpu_kuka import check_planes, make
from test_gpu_kuka_solver_detail import run_pair, table

pytestmark = pytest.mark.gpu
TOL = 1e-4


def pressing_actions(T, n, seed):
    rs = np.random.RandomState(seed)
    a = rs.randint(6, size=(T, n)).astype(np.int32)
    a[rs.rand(T, n) < 0.25] = 4
    return a


def test_forced_on_a_small_batch_step_by_step_and_fused(monkeypatch):
    monkeypatch.setenv("SRLHIP_KUKA_OCC", "1")
    n, T, seed0 = 256, 1010, 3
    actions = pressing_actions(T, n, 5)
    ora = kuka_clib.rollout(seed0 + np.arange(n), T, actions=actions, aux=True)
    assert ora["rows"][:, :, 0].sum() > 100                          # contact rows: the turn-taking generic path runs
    h = make(n, seed0=seed0)
    obs0 = h.reset()
    out = h.rollout(T, actions=actions)
    check_planes(ora, obs0, out)
    f = ora["final_state"]
    q = np.concatenate([h.get_state(_lib.F_KUKA_Q).T, h.get_state(_lib.F_KUKA_GRIPPER_Q).T], axis=1)
    assert np.abs(q - np.concatenate([f[:, :7], f[:, 30:35]], axis=1)).max() <= TOL
    h.close()
    h = make(n, seed0=seed0)                                          # the same, one launch per step
    assert np.abs(h.reset() - ora["obs0"]).max() <= TOL
    worst = 0.0
    for t in range(400):
        o, r, d = h.step(actions[t])
        assert np.array_equal(d, ora["done"][t]) and np.array_equal(r, ora["reward"][t]), t
        if not d.any():
            worst = max(worst, np.abs(h.get_state(_lib.F_KUKA_Q).T - ora["q"][t]).max())
    assert worst <= TOL
    h.close()


def test_both_variants_agree_on_state_handover(monkeypatch):
    """Half a rollout under one kernel, the second half under the other, on the same handle: the state planes are the interface."""
    n, T = 192, 600
    actions = pressing_actions(T, n, 9)
    ora = kuka_clib.rollout(np.arange(n), T, actions=actions)
    for first, second in (("0", "1"), ("1", "0")):
        h = make(n)
        monkeypatch.setenv("SRLHIP_KUKA_OCC", first)
        obs0 = h.reset()
        a = h.rollout(T // 2, actions=actions[:T // 2])
        monkeypatch.setenv("SRLHIP_KUKA_OCC", second)
        b = h.rollout(T - T // 2, actions=actions[T // 2:])
        out = {k: np.concatenate([a[k], b[k]]) for k in ("obs", "reward", "done")}
        check_planes(ora, obs0, out)
        h.close()


@pytest.mark.parametrize("kw", [dict(), dict(random_target=1)])
def test_default_dispatch_at_65536_envs(kw):
    """No override (ragged: 65536 + 37 envs, the last workgroup partly idle).  random_target = 1: the two-wavefront kernel (default from
    65536 envs for configurations the configuration-specialised instantiation does not cover); the reference's default configuration:
    that instantiation, one wavefront per SIMD, which round 5 measured faster at every batch size."""
    n, T = 65536 + 37, 100
    actions = pressing_actions(T, n, 6)
    h = make(n, **kw)
    obs0 = h.reset()
    out = h.rollout(T, actions=actions)
    ora = kuka_clib.rollout(np.arange(n), T, actions=actions, trace=False, **kw)
    check_planes(ora, obs0, out)
    f = ora["final_state"]
    assert np.abs(h.get_state(_lib.F_KUKA_Q).T - f[:, 0:7]).max() <= TOL
    assert np.array_equal(h.get_state(_lib.F_STEP_COUNT), f[:, 19].astype(np.int32))
    ret, length, fin = h.episode_stats()
    assert np.array_equal(length, ora["ep_stats"][:, 1].astype(np.int32)) and np.array_equal(ret, ora["ep_stats"][:, 0])
    h.close()


@pytest.mark.parametrize("kw", [dict(is_discrete=0), dict(is_discrete=0, rng_mode=_lib.RNG_PHILOX, agent=True),
                                dict(force_down=0, random_target=1), dict(variant=_lib.ENV_KUKA_MOVING)])
def test_other_modes_forced(monkeypatch, kw):
    monkeypatch.setenv("SRLHIP_KUKA_OCC", "1")
    kw = dict(kw)
    n, T, seed0 = 128, 700, 11
    agent, variant = kw.pop("agent", False), kw.pop("variant", _lib.ENV_KUKA_BUTTON)
    cfg = _lib.default_config(variant)
    cfg.num_envs, cfg.seed0 = n, seed0
    for k, v in kw.items():
        setattr(cfg, k, v)
    if cfg.is_discrete:
        actions = pressing_actions(T, n, 13)
    else:
        actions = np.random.RandomState(13).uniform(-1, 1, size=(T, n, 3)).astype(np.float32)
        actions[:, :, 2] -= 0.5                                       # mostly downwards: contacts
    okw = dict(random_target=bool(cfg.random_target), force_down=bool(cfg.force_down), is_discrete=bool(cfg.is_discrete), rng_mode=cfg.rng_mode,
               shape_reward=bool(cfg.shape_reward))
    h = _lib.Handle(cfg)
    obs0 = h.reset()
    kuka_clib.set_moving(variant == _lib.ENV_KUKA_MOVING)
    try:
        if agent:
            out = h.rollout(T)
            ora = kuka_clib.rollout(seed0 + np.arange(n), T, actions=None, trace=False, **okw)
            assert np.array_equal(ora["actions"], out["actions"])
        else:
            out = h.rollout(T, actions=actions)
            ora = kuka_clib.rollout(seed0 + np.arange(n), T, actions=actions, trace=False, **okw)
    finally:
        kuka_clib.set_moving(False)
    check_planes(ora, obs0, out, flags_exact=not cfg.shape_reward)
    h.close()


@pytest.mark.parametrize("detail", [0, 7])
def test_limit_rows_row_budget_and_detail_bits_forced(monkeypatch, detail):
    """The generic path's LDS loop in the shared work area: limit rows + contact rows + a row budget that overflows, under the
    default solver and under all three detail bits."""
    monkeypatch.setenv("SRLHIP_KUKA_OCC", "1")
    base, ora, err = run_pair(table(detail, tighten=0.3, budget=3), 256, 600, 17, random_target=1)
    lim, normals = ora["rows"][:, :, 1] // 1000, ora["rows"][:, :, 0]
    assert (lim > 0).mean() > 0.05 and ((lim > 0) & (normals > 0)).sum() > 5
