"""Edge cases of the C-ABI path on the GPU: ragged env counts (not a multiple of the 64-lane wavefront), single env,
masked resets, degenerate rollouts, out-of-range actions, a large handle."""
import numpy as np
import pytest

from oracle import clib, kuka_clib, mobile_oracle
from srlhip import _lib

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [1, 63, 65, 100, 1000])
def test_ragged_env_counts_mobile_bit_exact(n):
    T = 270
    actions = np.random.RandomState(n).randint(4, size=(T, n)).astype(np.int32)
    cfg = _lib.default_config(_lib.ENV_MOBILE)
    cfg.num_envs, cfg.seed0 = n, 11
    h = _lib.Handle(cfg)
    obs0 = h.reset()
    out = h.rollout(T, actions=actions)
    ora = clib.mobile_rollout(mobile_oracle.MOBILE, 11 + np.arange(n), T, actions=actions)
    assert np.array_equal(obs0, ora["obs0"]) and np.array_equal(out["obs"], ora["obs"])
    assert np.array_equal(out["reward"], ora["reward"]) and np.array_equal(out["done"], ora["done"])
    assert out["done"][250].all() and out["done"].sum() == n      # exactly one 251-step episode each
    h.close()


@pytest.mark.parametrize("n", [1, 65, 100])
def test_ragged_env_counts_kuka(n):
    T = 120
    actions = np.random.RandomState(n).randint(6, size=(T, n)).astype(np.int32)
    cfg = _lib.default_config(_lib.ENV_KUKA_BUTTON)
    cfg.num_envs, cfg.seed0 = n, 21
    h = _lib.Handle(cfg)
    obs0 = h.reset()
    out = h.rollout(T, actions=actions)
    ora = kuka_clib.rollout(21 + np.arange(n), T, actions=actions, trace=False)
    assert np.abs(ora["obs0"] - obs0).max() <= 1e-4 and np.abs(ora["obs"] - out["obs"]).max() <= 1e-4
    assert np.array_equal(ora["reward"], out["reward"]) and np.array_equal(ora["done"], out["done"])
    h.close()


def test_masked_reset_touches_only_selected_envs():
    n = 130
    cfg = _lib.default_config(_lib.ENV_MOBILE)
    cfg.num_envs, cfg.seed0 = n, 3
    h = _lib.Handle(cfg)
    h.reset()
    for _ in range(7):
        h.step(np.full(n, 1, np.int32))
    before = np.stack([h.get_state(_lib.F_POS_X), h.get_state(_lib.F_POS_Y)]), h.get_state(_lib.F_STEP_COUNT).copy()
    mask = (np.arange(n) % 3 == 0).astype(np.uint8)
    sentinel = np.full((n, 2), -77.0, np.float32)
    obs = h.reset(mask=mask, obs_out=sentinel)
    after = np.stack([h.get_state(_lib.F_POS_X), h.get_state(_lib.F_POS_Y)]), h.get_state(_lib.F_STEP_COUNT)
    keep = mask == 0
    assert np.array_equal(before[0][:, keep], after[0][:, keep]) and np.array_equal(before[1][keep], after[1][keep])
    assert (after[1][~keep] == 0).all() and (after[1][keep] == 7).all()
    assert (obs[keep] == -77.0).all() and (obs[~keep] != -77.0).all()      # unselected rows keep the caller's contents
    h.reset(mask=np.zeros(n, np.uint8))                                     # empty selection: a no-op
    assert np.array_equal(h.get_state(_lib.F_STEP_COUNT), after[1])
    h.close()


def test_degenerate_calls_are_rejected_not_executed():
    cfg = _lib.default_config(_lib.ENV_KUKA_BUTTON)
    cfg.num_envs = 4
    h = _lib.Handle(cfg)
    h.reset()
    with pytest.raises(_lib.SrlHipError):
        h.rollout(0)                                            # T = 0
    with pytest.raises(AssertionError):
        h.step(np.zeros(5, np.int32))                           # wrong batch size
    for bad in (dict(num_envs=0), dict(num_envs=-3), dict(action_repeat=0), dict(img_h=0, obs_mode=_lib.OBS_RAW_PIXELS)):
        c = _lib.default_config(_lib.ENV_KUKA_BUTTON)
        for k, v in bad.items():
            setattr(c, k, v)
        with pytest.raises(_lib.SrlHipError):
            _lib.Handle(c)
    from srlhip.vec_env import HipVecEnv
    env = HipVecEnv("KukaButtonGymEnv-v0", 4, env_kwargs={"srl_model": "ground_truth"})
    env.reset()
    with pytest.raises(IndexError):
        env.step([0, 1, 6, 2])                                  # the reference indexes a 6-entry list
    env.step([0, None, 5, 2])
    env.close(); h.close()
    # non-finite continuous actions / host noise are refused at the boundary (the full-model kernels are built with -fno-honor-nans)
    c = _lib.default_config(_lib.ENV_KUKA_BUTTON)
    c.num_envs, c.is_discrete = 4, 0
    h = _lib.Handle(c)
    h.reset()
    a = np.zeros((4, 3), np.float32)
    h.step(a)
    for bad in (np.nan, np.inf, -np.inf):
        a[2, 1] = bad
        with pytest.raises(_lib.SrlHipError):
            h.step(a)
        with pytest.raises(_lib.SrlHipError):
            h.rollout(3, actions=np.stack([a, a, a]))
    a[2, 1] = 0.5
    q = h.get_state(_lib.F_KUKA_Q)
    h.step(a)
    assert np.isfinite(h.get_state(_lib.F_KUKA_Q)).all() and not np.array_equal(q, h.get_state(_lib.F_KUKA_Q))
    h.close()


def test_large_handle_mobile_one_million_envs():
    n, T = 1 << 20, 4
    cfg = _lib.default_config(_lib.ENV_MOBILE)
    cfg.num_envs, cfg.seed0, cfg.rng_mode = n, 0, _lib.RNG_PHILOX
    h = _lib.Handle(cfg)
    obs0 = h.reset()
    out = h.rollout(T, want=("obs", "reward", "done", "actions"))
    assert obs0.shape == (n, 2) and out["obs"].shape == (T, n, 2) and np.isfinite(out["obs"]).all() and not out["done"].any()
    # the first and the last 64 envs of the handle follow the oracle exactly (global env id seeding, Philox streams)
    for lo in (0, n - 64):
        ora = clib.mobile_rollout(mobile_oracle.MOBILE, lo + np.arange(64), T, actions=None, rng_mode=clib.RNG_PHILOX)
        assert np.array_equal(obs0[lo:lo + 64], ora["obs0"]) and np.array_equal(out["obs"][:, lo:lo + 64], ora["obs"])
        assert np.array_equal(out["actions"][:, lo:lo + 64], ora["actions"])
    assert (np.abs(out["obs"]) < 10).all()
    h.close()


def test_sparse_reward_threshold_is_exact_at_the_ulp_boundary():
    """The kernel decides `norm <= 0.4` on the squared norm (exact threshold, no square root); walk the robot's x through
    every double within +-8 ulp of the boundary and compare with the oracle's sqrt(ddot) predicate."""
    cfg = _lib.default_config(_lib.ENV_MOBILE)
    n = 64
    cfg.num_envs, cfg.seed0 = n, 0
    h = _lib.Handle(cfg)
    h.reset()
    tx, ty = 3.6, 3.0                                        # default target
    checked = flips = 0
    for base_dx, y in ((0.4, ty), (0.24, ty - 0.32), (0.32, ty + 0.24), (0.4, ty + 1e-9)):   # all clear of the walls
        x0 = tx - base_dx - 0.1                              # action 1 moves +0.1 in x (dv = 0.1 exactly)
        xs = np.full(n, x0)
        for k in range(n):
            v = x0
            for _ in range(abs(k - n // 2)):
                v = np.nextafter(v, np.inf if k > n // 2 else -np.inf)
            xs[k] = v
        h.set_state(_lib.F_POS_X, xs); h.set_state(_lib.F_POS_Y, np.full(n, y))
        h.set_state(_lib.F_STEP_COUNT, np.zeros(n, np.int32))
        obs, rew, done = h.step(np.full(n, 1, np.int32))
        xn, yn = h.get_state(_lib.F_POS_X), h.get_state(_lib.F_POS_Y)
        want = np.array([1.0 if mobile_oracle.norm2(tx - a, ty - b) <= 0.4 else 0.0 for a, b in zip(xn, yn)], np.float32)
        assert np.array_equal(rew, want), (base_dx, rew, want)
        checked += n
        flips += int(want.min() != want.max())
    assert checked == 4 * n and flips >= 2                   # the scan really straddles the boundary
    h.close()
