"""GPU parity, KukaButtonGymEnv: HIP stepper (through the C-ABI) vs the plain-C
oracle on identical seeds/actions.  North-star bar: joint positions within 1e-4,
discrete reward / done flags bit-exact."""
import numpy as np
import pytest

from oracle import clib, kuka_clib
from srlhip import _lib, kuka_model

pytestmark = pytest.mark.gpu
TOL = 1e-4


def make(n, rng_mode=_lib.RNG_MT19937, auto_reset=1, seed0=0, first=0, **kw):
    cfg = _lib.default_config(_lib.ENV_KUKA_BUTTON)           # kuka_model = FULL: the tree lane-group kernel (conftest puts the oracle in the same mode)
    cfg.num_envs, cfg.rng_mode, cfg.auto_reset, cfg.seed0, cfg.first_env_id = n, rng_mode, auto_reset, seed0, first
    for k, v in kw.items():
        setattr(cfg, k, v)
    return _lib.Handle(cfg)


def check_planes(ora, obs0, out, flags_exact=True):
    assert np.abs(ora["obs0"] - obs0).max() <= TOL
    assert np.abs(ora["obs"] - out["obs"]).max() <= TOL
    assert np.array_equal(ora["done"], out["done"])
    if flags_exact:
        assert np.array_equal(ora["reward"], out["reward"])
    else:
        assert np.abs(ora["reward"] - out["reward"]).max() <= TOL


def test_step_by_step_trajectories_256_envs():
    """Per-step launches over 1001+ steps; joint positions checked after EVERY step."""
    n, T, seed0 = 256, 1010, 3
    actions = np.random.RandomState(5).randint(6, size=(T, n)).astype(np.int32)
    h = make(n, seed0=seed0)
    obs0 = h.reset()
    ora = kuka_clib.rollout(seed0 + np.arange(n), T, actions=actions)
    assert np.abs(ora["obs0"] - obs0).max() <= TOL
    worst = 0.0
    for t in range(T):
        o, r, d = h.step(actions[t])
        assert np.array_equal(d, ora["done"][t]), t
        assert np.array_equal(r, ora["reward"][t]), t
        assert np.abs(o - ora["obs"][t]).max() <= TOL, t
        if not d.any():                       # q of a just-reset env belongs to the next episode
            worst = max(worst, np.abs(h.get_state(_lib.F_KUKA_Q).T - ora["q"][t]).max())
    assert worst <= TOL
    print("max |q_gpu - q_oracle| =", worst)
    ret, length, fin = h.episode_stats()
    assert np.array_equal(length, ora["ep_stats"][:, 1].astype(np.int32))
    assert np.array_equal(fin, ora["ep_stats"][:, 2].astype(np.int32))
    assert np.array_equal(ret, ora["ep_stats"][:, 0])
    assert fin.min() >= 1
    h.close()


def test_fused_rollout_4096_envs():
    """BASELINE config 3 size: 4096 envs, one fused launch, every env crosses >= 1 auto-reset."""
    n, T = 4096, 1001
    actions = np.random.RandomState(6).randint(6, size=(T, n)).astype(np.int32)
    # bias towards 'down' so that most episodes end by button/table contact well before 1001 steps
    down = np.random.RandomState(7).rand(T, n) < 0.25
    actions[down] = 4
    h = make(n)
    obs0 = h.reset()
    out = h.rollout(T, actions=actions)
    kuka_clib.margins_reset()
    ora = kuka_clib.rollout(np.arange(n), T, actions=actions, trace=False)
    check_planes(ora, obs0, out)
    # Flag parity is bit-exact because no threshold is approached to within the steppers' numerical difference (~1e-11: the same
    # sums, associated differently; the predicates themselves are evaluated unfused on both sides): the smallest
    # |value - threshold| behind a contact / distance flag over these 4.1e6 env-steps, recorded by the oracle
    m = kuka_clib.margins()
    print("flag margins over {} env-steps: {}".format(n * T, {k: float(v) for k, v in m.items()}))
    assert min(m.values()) > 1e-10
    f = ora["final_state"]
    assert np.abs(h.get_state(_lib.F_KUKA_Q).T - f[:, 0:7]).max() <= TOL
    assert np.abs(h.get_state(_lib.F_KUKA_QD).T - f[:, 7:14]).max() <= 1e-3
    assert np.array_equal(h.get_state(_lib.F_STEP_COUNT), f[:, 19].astype(np.int32))
    assert np.array_equal(h.get_state(_lib.F_KUKA_COUNTERS).T, f[:, 20:23].astype(np.int32))
    ret, length, fin = h.episode_stats()
    assert np.array_equal(fin, ora["ep_stats"][:, 2].astype(np.int32)) and fin.min() >= 1
    assert np.array_equal(length, ora["ep_stats"][:, 1].astype(np.int32))
    h.close()


@pytest.mark.parametrize("kw", [
    dict(random_target=1, shape_reward=1),
    dict(action_repeat=3, force_down=0, max_distance=0.28),
    dict(obs_mode=_lib.OBS_JOINTS_POSITION),
    dict(obs_mode=_lib.OBS_JOINTS),
])
def test_env_options(kw):
    n, T = 128, 600
    actions = np.random.RandomState(8).randint(-1, 6, size=(T, n)).astype(np.int32)      # -1 == None action
    h = make(n, seed0=40, **kw)
    obs0 = h.reset()
    out = h.rollout(T, actions=actions)
    okw = {k: v for k, v in kw.items()}
    ora = kuka_clib.rollout(40 + np.arange(n), T, actions=actions, trace=False, **okw)
    check_planes(ora, obs0, out, flags_exact=not kw.get("shape_reward"))
    h.close()


@pytest.mark.parametrize("joints", [0, 1])
def test_continuous_actions(joints):
    n, T, adim = 64, 400, 7 if joints else 3
    actions = np.random.RandomState(9).uniform(-1, 1, size=(T, n, adim)).astype(np.float32)
    h = make(n, seed0=9, is_discrete=0, action_joints=joints)
    obs0 = h.reset()
    out = h.rollout(T, actions=actions)
    ora = kuka_clib.rollout(9 + np.arange(n), T, actions=actions, is_discrete=False, action_joints=bool(joints), trace=False)
    check_planes(ora, obs0, out)
    h.close()


def test_philox_random_agent_rollout():
    """Throughput mode (what bench.py times): Philox env stream + device-sampled actions."""
    n, T = 1024, 500
    h = make(n, rng_mode=_lib.RNG_PHILOX, seed0=2)
    obs0 = h.reset()
    out = h.rollout(T)
    ora = kuka_clib.rollout(2 + np.arange(n), T, actions=None, rng_mode=kuka_clib.RNG_PHILOX, trace=False)
    assert np.array_equal(ora["actions"], out["actions"])
    check_planes(ora, obs0, out)
    h.close()


def test_host_rng_mode_no_auto_reset():
    """RNG_HOST harness: caller supplies np_random's draws; finished envs reset via srlhip_reset(mask)."""
    from oracle import gym_seeding
    n, T = 32, 700
    rngs = [gym_seeding.np_random(500 + i)[0] for i in range(n)]
    actions = np.random.RandomState(11).randint(6, size=(T, n)).astype(np.int32)
    actions[np.random.RandomState(12).rand(T, n) < 0.3] = 4
    h = make(n, rng_mode=_lib.RNG_HOST, auto_reset=0)

    def reset_draws(r):
        out = []
        for _ in range(5):
            out += [r.rand(), float(r.randint(3))]
        return out

    obs = h.reset(host_rand=np.array([reset_draws(r) for r in rngs]))
    ora = kuka_clib.rollout(500 + np.arange(n), T, actions=actions)
    assert np.abs(obs - ora["obs0"]).max() <= TOL
    for t in range(T):
        noise = np.array([r.normal(0.0, scale=0.01) for r in rngs])
        o, r_, d = h.step(actions[t], host_noise=noise)
        assert np.array_equal(d, ora["done"][t]) and np.array_equal(r_, ora["reward"][t])
        if d.any():
            rand = np.zeros((n, 10))
            for i in np.nonzero(d)[0]:
                rand[i] = reset_draws(rngs[i])
            o = h.reset(mask=d, host_rand=rand, obs_out=o.copy())
        assert np.abs(o - ora["obs"][t]).max() <= TOL
    h.close()


def test_masked_resets_under_mt19937_do_not_touch_running_streams():
    """environments/dataset_generator.py's pattern (rng_mode MT19937, auto_reset off, srlhip_reset(mask) for the finished envs) over
    enough steps that every env's generator twists at least twice: a masked-out env must neither draw nor regenerate its state block
    (round-3 advisor finding: kuka_tree_reset_k used to run the reset draws of masked-out rows, and GroupMt's in-place twist then left
    a running env with a fresh block and a stale index)."""
    n, T = 37, 900                      # ragged: the last wavefront carries shadow rows too
    actions = np.random.RandomState(21).randint(6, size=(T, n)).astype(np.int32)
    actions[np.random.RandomState(22).rand(T, n) < 0.3] = 4
    h = make(n, auto_reset=0, seed0=700)
    obs = h.reset()
    ora = kuka_clib.rollout(700 + np.arange(n), T, actions=actions)
    assert np.abs(obs - ora["obs0"]).max() <= TOL
    n_masked = 0
    for t in range(T):
        o, r_, d = h.step(actions[t])
        assert np.array_equal(d, ora["done"][t]) and np.array_equal(r_, ora["reward"][t]), t
        if d.any():
            n_masked += 1
            o = h.reset(mask=d, obs_out=o.copy())
        assert np.abs(o - ora["obs"][t]).max() <= TOL, t
    assert n_masked >= 20 and ora["ep_stats"][:, 2].min() >= 1
    h.close()


def test_sharding_invariance():
    n, T = 256, 300
    actions = np.random.RandomState(13).randint(6, size=(T, n)).astype(np.int32)
    full = make(n, seed0=21, random_target=1)
    o_full = full.reset()
    r_full = full.rollout(T, actions=actions)
    for g in range(2):
        hs = make(n // 2, seed0=21, first=g * n // 2, random_target=1)
        sl = slice(g * n // 2, (g + 1) * n // 2)
        assert np.array_equal(hs.reset(), o_full[sl])
        r = hs.rollout(T, actions=np.ascontiguousarray(actions[:, sl]))
        for k in ("obs", "reward", "done"):
            assert np.array_equal(r[k], r_full[k][:, sl])
        hs.close()
    full.close()


def test_state_checkpoint_restore_and_error_paths():
    """get_state/set_state round trip (env-state snapshot), and misuse that must fail loudly."""
    n, T = 64, 60
    actions = np.random.RandomState(21).randint(6, size=(2 * T, n)).astype(np.int32)
    a = make(n, rng_mode=_lib.RNG_PHILOX, seed0=5)
    a.reset()
    a.rollout(T, actions=actions[:T])
    snap = {f: a.get_state(f) for f in (_lib.F_KUKA_Q, _lib.F_KUKA_QD, _lib.F_KUKA_EE_TARGET, _lib.F_KUKA_BUTTON_Q)}
    ref = a.rollout(T, actions=actions[T:])
    b = make(n, rng_mode=_lib.RNG_PHILOX, seed0=5)
    b.reset()
    b.rollout(T, actions=actions[:T])                        # same RNG/counter state as `a` at the snapshot
    b.set_state(_lib.F_KUKA_QD, np.zeros((7, n)))            # perturb, then restore from the snapshot
    b.set_state(_lib.F_KUKA_Q, snap[_lib.F_KUKA_Q] + 0.01)
    for f, v in snap.items():
        b.set_state(f, v)
    out = b.rollout(T, actions=actions[T:])
    assert np.array_equal(out["obs"], ref["obs"]) and np.array_equal(out["reward"], ref["reward"])
    with pytest.raises(_lib.SrlHipError):
        b.get_state(_lib.F_POS_X)                             # a MobileRobot field on a Kuka handle
    hst = make(4, rng_mode=_lib.RNG_HOST, auto_reset=0)
    with pytest.raises(_lib.SrlHipError):
        hst.rollout(4)                                        # fused rollout needs a device RNG
    with pytest.raises(_lib.SrlHipError):
        hst.reset()                                           # RNG_HOST reset without the draws
    a.close(); b.close(); hst.close()


def test_moving_button_env():
    """KukaMovingButtonGymEnv-v0 on the GPU vs the oracle (whose wrapper logic is pinned to the reference source)."""
    n, T = 128, 1600
    actions = np.random.RandomState(31).randint(6, size=(T, n)).astype(np.int32)
    actions[:, :8] = -1                                                 # eight idle envs (None action) run into the 1500-step limit
    cfg = _lib.default_config(_lib.ENV_KUKA_MOVING)
    cfg.num_envs, cfg.seed0, cfg.shape_reward = n, 12, 1
    h = _lib.Handle(cfg)
    obs0 = h.reset()
    out = h.rollout(T, actions=actions)
    kuka_clib.set_moving(True)
    try:
        ora = kuka_clib.rollout(12 + np.arange(n), T, actions=actions, shape_reward=True, trace=False)
    finally:
        kuka_clib.set_moving(False)
    check_planes(ora, obs0, out, flags_exact=False)
    ret, length, fin = h.episode_stats()
    assert np.array_equal(length, ora["ep_stats"][:, 1].astype(np.int32)) and length[:8].tolist() == [1501] * 8
    h.close()
    from environments.registry import registered_env
    env = registered_env["KukaMovingButtonGymEnv-v0"][0](srl_model="ground_truth")
    env.seed(1); env.reset()
    y0 = env.getTargetPos()[1]
    env.step(0)
    assert abs(abs(env.getTargetPos()[1] - y0) - 0.001) < 1e-12 and env.max_steps == 1500
    env.close()


def two_button_env(full):
    """Kuka2ButtonGymEnv-v0 on the GPU (NB = 2 kernels: second button body, goal switching) vs the oracle, whose wrapper
    logic is pinned to the reference source (tests/test_kuka_2button_golden.py).  Env 0 presses both buttons in order."""
    from test_kuka_kernel_source_on_host import two_button_actions
    from test_kuka_tree_kernel_source_on_host import two_button_actions_full
    n, T = 128, 1600
    actions = two_button_actions_full(n, T) if full else two_button_actions(n, T)
    for kw in (dict(), dict(shape_reward=1, random_target=1)):
        cfg = _lib.default_config(_lib.ENV_KUKA_2BUTTON)
        assert cfg.max_distance == 2.0 and cfg.force_down == 0 and cfg.kuka_model == _lib.KUKA_MODEL_FULL
        cfg.num_envs, cfg.seed0 = n, 60
        if not full:
            cfg.kuka_model = _lib.KUKA_MODEL_LUMPED
        for k, v in kw.items():
            setattr(cfg, k, v)
        h = _lib.Handle(cfg)
        assert h.kuka_kernel() == ("tree" if full else "lane") and kuka_clib.is_full() == full
        obs0 = h.reset()
        b2_xy0 = h.get_state(_lib.F_KUKA_BUTTON2_XY).copy()
        out = h.rollout(T, actions=actions)
        kuka_clib.set_variant(kuka_clib.VARIANT_TWO)
        try:
            ora = kuka_clib.rollout(60 + np.arange(n), T, actions=actions, force_down=False, max_distance=2.0,
                                    shape_reward=bool(kw.get("shape_reward")), random_target=bool(kw.get("random_target")), trace=False)
        finally:
            kuka_clib.set_variant(kuka_clib.VARIANT_BUTTON)
        check_planes(ora, obs0, out, flags_exact=not kw)
        ret, length, fin = h.episode_stats()
        assert np.array_equal(length, ora["ep_stats"][:, 1].astype(np.int32)) and np.array_equal(fin, ora["ep_stats"][:, 2].astype(np.int32))
        fs = ora["final_state"]
        goal = h.get_state(_lib.F_KUKA_GOAL)
        assert np.array_equal(goal[0], fs[:, 26].astype(np.int32)) and np.array_equal(goal[1], fs[:, 27].astype(np.int32))
        assert np.abs(h.get_state(_lib.F_KUKA_BUTTON2_Q).T - fs[:, 24:26]).max() <= TOL
        assert np.array_equal(h.get_state(_lib.F_KUKA_BUTTON2_XY).T, fs[:, 28:30])           # reset draws: bit-exact
        if not kw:
            assert (b2_xy0.T == [0.5, -0.125]).all()
            first = int(np.argmax(out["done"][:, 0]))
            assert out["done"][first, 0] and out["reward"][first, 0] == 1.0 and out["reward"][:first, 0].sum() == 4
            assert length[1] == 1501                                                          # idle env: counter > 1500
        h.close()


def test_two_button_env():
    """the default: the full arm + gripper model on the tree lane-group kernel's two-button form"""
    two_button_env(True)
    from environments.registry import registered_env
    env = registered_env["Kuka2ButtonGymEnv-v0"][0](srl_model="ground_truth")
    env.seed(1)
    o = env.reset()
    assert env.max_steps == 1500 and env.goal_id == 0 and env.n_contacts == [0, 0] and o.shape == (3,)
    assert np.allclose(env.button_all_pos[0], [0.5, 0.125, 0.08]) and np.allclose(env.button_all_pos[1], [0.5, -0.125, 0.08])
    assert np.array_equal(env.getTargetPos(), env.button_all_pos[0])
    env.close()


@pytest.mark.lumped_kuka
def test_two_button_env_lumped_model():
    """rounds 1-2: the lumped-gripper model on the lane-per-env kernel (kuka_model = LUMPED)"""
    two_button_env(False)


def test_rand_button_env():
    """KukaRandButtonGymEnv-v0 on the GPU vs the oracle (reset draws pinned to the reference source,
    tests/test_kuka_rand_button_golden.py); distractor positions and keep flags are bit-exact.  Round 4: the ten objects and the
    kicked ball are free bodies (kuka_rand_button_gym_env.py:59-71,111-125) — arm-sphere contacts, table contact + friction, the
    step-10 kick: object poses after the 1100 steps (and at a moment when the balls are still rolling) against the oracle."""
    n, T = 128, 1100
    rs = np.random.RandomState(41)
    actions = rs.randint(6, size=(T, n)).astype(np.int32)
    actions[rs.rand(T, n) < 0.35] = 4                          # go down often: the gripper reaches the objects lying around the button
    cfg = _lib.default_config(_lib.ENV_KUKA_RAND)
    cfg.num_envs, cfg.seed0, cfg.random_target = n, 80, 1
    h = _lib.Handle(cfg)
    obs0 = h.reset()
    objs = h.get_state(_lib.F_KUKA_OBJECTS).T.reshape(n, 10, 3).copy()
    bxy = h.get_state(_lib.F_KUKA_BUTTON_XY).T
    keep = (objs[:, :, 0] < bxy[:, None, 0] - 0.1) | (objs[:, :, 0] > bxy[:, None, 0] + 0.1) | \
           (objs[:, :, 1] < bxy[:, None, 1] - 0.1) | (objs[:, :, 1] > bxy[:, None, 1] + 0.1)
    assert np.array_equal(objs[:, :, 2] > 0, keep) and (np.abs(objs[:, :, 0] - 0.5) <= 0.15).all() and (np.abs(objs[:, :, 1]) <= 0.3).all()
    bodies0 = h.get_state(_lib.F_KUKA_BODIES).T.reshape(n, 11, 6)
    assert np.array_equal(bodies0[:, :10, 0], objs[:, :, 0]) and np.array_equal(bodies0[:, :10, 1], objs[:, :, 1])      # at rest where they were drawn
    assert (bodies0[:, :, 3:] == 0).all() and np.allclose(bodies0[:, 10, :3], [0.25, -0.2, -0.195 + 0.03])
    kuka_clib.set_variant(kuka_clib.VARIANT_RAND)
    try:
        tr = kuka_clib.command_trace(80, 1, np.zeros(1, np.int32), random_target=True)
        assert np.array_equal(kuka_clib.last_objects(), objs[0])                            # env 0: same 20 draws
        # (1) 30 steps: the kick of step 10 has happened, the balls are on their way
        ob30 = kuka_clib.body_trace(n)
        ora30 = kuka_clib.rollout(80 + np.arange(n), 30, actions=actions[:30], random_target=True, trace=False)
        kuka_clib.body_trace_off()
        h30 = _lib.Handle(cfg)                                  # (a handle of its own: the main one keeps its RNG streams for the full run)
        h30.reset()
        out30 = h30.rollout(30, actions=actions[:30])
        b30 = h30.get_state(_lib.F_KUKA_BODIES).T.reshape(n, 11, 6)
        h30.close()
        assert np.abs(b30 - ob30[:, :, :6]).max() <= 1e-9
        assert np.abs(ob30[:, 10, 3:5]).max(axis=1).min() > 0.05 and (np.hypot(ob30[:, 10, 0] - 0.25, ob30[:, 10, 1] + 0.2) > 1e-3).all()   # every ball is on its way (first quadrant)
        assert (ob30[:, 10, 3:5] >= 0).all()
        assert np.array_equal(out30["reward"], ora30["reward"]) and np.array_equal(out30["done"], ora30["done"])
        # (2) the full run
        ob = kuka_clib.body_trace(n)
        ora = kuka_clib.rollout(80 + np.arange(n), T, actions=actions, random_target=True, trace=False, aux=True)
        kuka_clib.body_trace_off()
    finally:
        kuka_clib.set_variant(kuka_clib.VARIANT_BUTTON)
    out = h.rollout(T, actions=actions)
    check_planes(ora, obs0, out)
    bodies = h.get_state(_lib.F_KUKA_BODIES).T.reshape(n, 11, 6)
    on = ob[:, :, 6] > 0
    assert np.abs(bodies - ob[:, :, :6])[on].max() <= 1e-9                                   # object poses and velocities after 1100 steps
    assert ora["rows"][:, :, 2].sum() > 50                                                   # arm <-> body contact rows were part of it
    ret, length, fin = h.episode_stats()
    assert np.array_equal(length, ora["ep_stats"][:, 1].astype(np.int32)) and fin.min() >= 1
    h.close()


@pytest.mark.lumped_kuka
@pytest.mark.parametrize("kernel", ["lane", "group"])
def test_both_kernels_at_the_headline_size(kernel, monkeypatch):
    """Lumped-gripper model: the library picks the lane-group kernel (16 lanes per env, kuka_group.hpp) for batches up to 12288
    envs and the lane-per-env kernel (kuka_core.hpp) above; SRLHIP_KUKA_KERNEL forces one.  Both against the oracle at 4096
    envs, in the throughput mode (Philox streams, device-sampled actions), through two auto-resets per env."""
    monkeypatch.setenv("SRLHIP_KUKA_KERNEL", kernel)
    n, T = 4096, 1100
    h = make(n, rng_mode=_lib.RNG_PHILOX, seed0=11, kuka_model=_lib.KUKA_MODEL_LUMPED)
    assert h.kuka_kernel() == kernel
    obs0 = h.reset()
    out = h.rollout(T)
    ora = kuka_clib.rollout(11 + np.arange(n), T, actions=None, rng_mode=kuka_clib.RNG_PHILOX, trace=False)
    assert np.array_equal(ora["actions"], out["actions"])
    check_planes(ora, obs0, out)
    assert np.abs(h.get_state(_lib.F_KUKA_Q).T - ora["final_state"][:, :7]).max() <= TOL
    ret, length, fin = h.episode_stats()
    assert np.array_equal(fin, ora["ep_stats"][:, 2].astype(np.int32)) and fin.min() >= 1
    h.close()


@pytest.mark.lumped_kuka
def test_kernels_agree_on_state_handover():
    """Lumped model: a rollout may be continued by the other kernel — both read and write the same state planes."""
    import os
    n, T = 512, 300
    actions = np.random.RandomState(21).randint(6, size=(2 * T, n)).astype(np.int32)
    actions[np.random.RandomState(22).rand(2 * T, n) < 0.3] = 4
    outs = []
    for first, second in (("group", "lane"), ("lane", "group")):
        h = make(n, seed0=5, kuka_model=_lib.KUKA_MODEL_LUMPED)
        h.reset()
        os.environ["SRLHIP_KUKA_KERNEL"] = first
        a = h.rollout(T, actions=actions[:T])
        os.environ["SRLHIP_KUKA_KERNEL"] = second
        b = h.rollout(T, actions=actions[T:])
        os.environ.pop("SRLHIP_KUKA_KERNEL")
        outs.append((a, b, h.get_state(_lib.F_KUKA_Q)))
        h.close()
    ora = kuka_clib.rollout(5 + np.arange(n), 2 * T, actions=actions, trace=False)
    for a, b, q in outs:
        assert np.array_equal(np.concatenate([a["done"], b["done"]]), ora["done"])
        assert np.array_equal(np.concatenate([a["reward"], b["reward"]]), ora["reward"])
        assert np.abs(np.concatenate([a["obs"], b["obs"]]) - ora["obs"]).max() <= TOL
        assert np.abs(q.T - ora["final_state"][:, :7]).max() <= TOL


@pytest.mark.lumped_kuka
def test_runtime_model_table_on_the_device():
    """srlhip_set_kuka_model (lumped model): another arm (masses, centres of mass, inertias, link lengths, joint frames, damping, gripper
    geometry) installed as DATA — the lane-group kernel integrates it, the oracle with the same table agrees; the baked table
    sent through the same entry point reproduces the baked model."""
    from srlhip import kuka_model
    n, T = 256, 500
    actions = np.random.RandomState(31).randint(6, size=(T, n)).astype(np.int32)
    actions[np.random.RandomState(32).rand(T, n) < 0.3] = 4
    m0 = kuka_clib.get_model()
    rs = np.random.RandomState(5)
    m = {k: (np.array(v, copy=True) if not np.isscalar(v) else v) for k, v in m0.items()}
    m["mass"] = m0["mass"] * rs.uniform(0.8, 1.3, 7); m["com"] = m0["com"] + rs.uniform(-0.01, 0.01, (7, 3))
    m["inertia"] = m0["inertia"] * rs.uniform(0.8, 1.3, (7, 3)); m["joint_xyz"] = m0["joint_xyz"] * 1.05
    m["joint_rpy"] = m0["joint_rpy"] + rs.uniform(-0.03, 0.03, (7, 3)); m["joint_damping"] = 0.7
    m["gripper_point"] = m0["gripper_point"] + np.array([0.0, 0.01, 0.02]); m["sphere"] = m0["sphere"] * 1.1
    base = kuka_clib.rollout(7 + np.arange(n), T, actions=actions, trace=False)
    try:
        for table, ref in ((m0, base), (m, None)):
            h = make(n, seed0=7, kuka_model=_lib.KUKA_MODEL_LUMPED)
            h.set_kuka_model(kuka_model.to_table(table))
            assert h.kuka_kernel() == "group"
            obs0 = h.reset()
            out = h.rollout(T, actions=actions)
            if ref is None:
                kuka_clib.set_model(table)
                ref = kuka_clib.rollout(7 + np.arange(n), T, actions=actions, trace=False)
                assert np.abs(ref["obs"] - base["obs"]).max() > 1e-2          # it IS a different arm
            check_planes(ref, obs0, out)
            assert np.abs(h.get_state(_lib.F_KUKA_Q).T - ref["final_state"][:, :7]).max() <= TOL
            h.close()
    finally:
        kuka_clib.set_model(m0)
    # lumped Kuka2Button handles are stepped by the lane-per-env kernel (baked model only): they refuse a table
    cfg = _lib.default_config(_lib.ENV_KUKA_2BUTTON)
    cfg.num_envs, cfg.kuka_model = 4, _lib.KUKA_MODEL_LUMPED
    h = _lib.Handle(cfg)
    assert h.kuka_kernel() == "lane"
    with pytest.raises(_lib.SrlHipError):
        h.set_kuka_model(kuka_model.to_table(m0))
    h.close()


def test_full_model_is_the_default_and_gripper_state_matches_the_oracle():
    """KukaButtonGymEnv handles integrate the full 12-DoF gripper tree by default (tree lane-group kernel, every batch size):
    arm AND gripper joints against the oracle's full mode, contact / friction steps included; set_state(GRIPPER_Q) round trip."""
    n, T = 512, 700
    actions = np.random.RandomState(41).randint(6, size=(T, n)).astype(np.int32)
    actions[np.random.RandomState(42).rand(T, n) < 0.25] = 4            # press down often: contacts with friction rows
    h = make(n, seed0=9, random_target=1)
    assert h.cfg.kuka_model == _lib.KUKA_MODEL_FULL and h.kuka_kernel() == "tree" and kuka_clib.is_full()
    obs0 = h.reset()
    gq0 = h.get_state(_lib.F_KUKA_GRIPPER_Q)
    assert np.abs(gq0[1]).max() < 0.05 and np.abs(gq0[3]).max() < 0.05      # fingers closed by the 500 settle steps (they start at -+0.3)
    out = h.rollout(T, actions=actions)
    ora = kuka_clib.rollout(9 + np.arange(n), T, actions=actions, random_target=True, aux=True, trace=False)
    check_planes(ora, obs0, out)
    assert ora["rows"][:, :, 0].sum() > 50 and ora["rows"][:, :, 1].sum() > 50          # contact normals and friction rows were exercised
    fs = ora["final_state"]
    assert np.abs(h.get_state(_lib.F_KUKA_Q).T - fs[:, :7]).max() <= TOL
    assert np.abs(h.get_state(_lib.F_KUKA_GRIPPER_Q).T - fs[:, 30:35]).max() <= TOL
    assert np.abs(h.get_state(_lib.F_KUKA_GRIPPER_QD).T - fs[:, 35:40]).max() <= 1e-3
    g = h.get_state(_lib.F_KUKA_GRIPPER_Q).copy()
    grip_before = h.get_state(_lib.F_KUKA_GRIPPER).copy()
    g[1] += 0.2                                                          # open the left finger: getArmPos() is that finger's COM
    h.set_state(_lib.F_KUKA_GRIPPER_Q, g)
    assert np.abs(h.get_state(_lib.F_KUKA_GRIPPER) - grip_before).max() > 1e-3
    h.close()
    # a lumped handle of the same env: the rounds 1-2 kernels, a different trajectory
    h2 = make(64, seed0=9, random_target=1, kuka_model=_lib.KUKA_MODEL_LUMPED)
    assert h2.kuka_kernel() == "group"
    h2.reset()
    out2 = h2.rollout(100, actions=actions[:100, :64])
    assert np.abs(out2["obs"] - out["obs"][:100, :64]).max() > 1e-3
    h2.close()
    for kind in (_lib.ENV_KUKA_BUTTON, _lib.ENV_KUKA_MOVING, _lib.ENV_KUKA_2BUTTON, _lib.ENV_KUKA_RAND):
        assert _lib.default_config(kind).kuka_model == _lib.KUKA_MODEL_FULL          # every Kuka env


def test_full_model_runtime_table():
    """srlhip_set_kuka_tree_model: a perturbed gripper (masses, finger motor forces, sphere radii, friction) installed as data is
    integrated identically by the device and by the oracle with the same table."""
    n, T = 128, 500
    actions = np.random.RandomState(51).randint(6, size=(T, n)).astype(np.int32)
    actions[np.random.RandomState(52).rand(T, n) < 0.3] = 4
    t0 = _lib.kuka_tree_default_model()
    assert np.abs(t0 - kuka_clib.get_tree_model()).max() < 1e-15         # product table == oracle table
    t = t0.copy()
    J = 1 + 33 * np.arange(12)                                            # first double of each joint record
    t[J[8:] + 19] *= 1.5                                                  # masses of the finger / tip bodies
    t[J[8] + 30] = 3.0; t[J[10] + 30] = 1.5                               # finger motor forces
    S = 1 + 33 * 12 + 9 + 6 * np.arange(16)
    t[S + 4] *= 1.1; t[S + 5] *= 0.5                                      # sphere radii, friction coefficients
    try:
        h = make(n, seed0=13)
        h.set_kuka_tree_model(t)
        obs0 = h.reset()
        out = h.rollout(T, actions=actions)
        base = kuka_clib.rollout(13 + np.arange(n), T, actions=actions, trace=False)
        kuka_clib.set_tree_model(t)
        ref = kuka_clib.rollout(13 + np.arange(n), T, actions=actions, trace=False)
        assert np.abs(ref["obs"] - base["obs"]).max() > 1e-4              # it IS a different gripper
        check_planes(ref, obs0, out)
        assert np.abs(h.get_state(_lib.F_KUKA_GRIPPER_Q).T - ref["final_state"][:, 30:35]).max() <= TOL
        h.close()
    finally:
        kuka_clib.set_full(True)                                          # rebuilds the default table


def test_full_model_joint_limit_rows_and_row_budget():
    """The tree kernel's general path with joint-LIMIT rows (LDS coupling planes; a random agent never reaches the real limits):
    a table whose limits of joints 3 and 5 sit 0.3 rad around the settled pose and a row budget of 3 that fills up; limit rows,
    contact rows and both at once, several envs of a wavefront carrying different row sets at the same time."""
    n, T = 256, 600
    rs = np.random.RandomState(61)
    actions = rs.randint(6, size=(T, n)).astype(np.int32)
    actions[rs.rand(T, n) < 0.3] = 4
    t = _lib.kuka_tree_default_model().copy()
    J = 1 + 33 * np.arange(12)
    q_settled = np.array([0.0, 0.6, 0.0, -0.86, 0.0, 1.68, 0.0])
    jj = np.array([3, 5])
    t[J[jj] + 16] = q_settled[jj] - 0.3
    t[J[jj] + 17] = q_settled[jj] + 0.3
    t[kuka_model.TREE_MAX_GENERIC_ROWS] = 3.0
    try:
        h = make(n, seed0=17, random_target=1)
        h.set_kuka_tree_model(t)
        obs0 = h.reset()
        out = h.rollout(T, actions=actions)
        kuka_clib.set_tree_model(t)
        ora = kuka_clib.rollout(17 + np.arange(n), T, actions=actions, random_target=True, aux=True, trace=False)
        lim, normals = ora["rows"][:, :, 1] // 1000, ora["rows"][:, :, 0]
        # limit rows on > 5 % of the env-steps, contact rows beside them, and the budget of 3 reached
        assert (lim > 0).mean() > 0.05 and ((lim > 0) & (normals > 0)).sum() > 20 and ((lim + normals) >= 3).sum() > 5
        check_planes(ora, obs0, out)
        assert np.abs(h.get_state(_lib.F_KUKA_Q).T - ora["final_state"][:, :7]).max() <= TOL
        h.close()
    finally:
        kuka_clib.set_full(True)


@pytest.mark.parametrize("n", [1, 5, 7, 130])
def test_ragged_env_counts_with_mt19937_streams(n):
    """env counts that do not fill the last wavefront (its spare lane groups shadow the last env, MT19937 state words included:
    the lane-group generator fetches and twists them across the 16 lanes) through several episodes and twists"""
    T = 1300
    actions = np.random.RandomState(3).randint(6, size=(T, n)).astype(np.int32)
    h = make(n, seed0=21, random_target=1)
    assert h.cfg.rng_mode == _lib.RNG_MT19937
    obs0 = h.reset()
    out = h.rollout(T, actions=actions)
    ora = kuka_clib.rollout(21 + np.arange(n), T, actions=actions, random_target=True, trace=False)
    check_planes(ora, obs0, out)
    assert ora["done"].sum() >= 2
    h.close()


@pytest.mark.parametrize("mode", ["continuous_philox_agent", "joints_mt19937", "host_rng_masked_resets"])
def test_rand_button_env_other_action_modes_and_ragged_batches(mode):
    """KukaRandButtonGymEnv's free bodies under the other kernel instantiations: continuous Cartesian actions sampled on the device
    (Philox agent), joint-space actions with the reference-exact MT19937 streams, and the host-RNG harness with masked resets — all on
    batches that do not fill the last wavefront (its spare lane groups shadow the last env, bodies included)."""
    cfg = _lib.default_config(_lib.ENV_KUKA_RAND)
    T = 700
    kuka_clib.set_variant(kuka_clib.VARIANT_RAND)
    try:
        if mode == "continuous_philox_agent":
            n = 7
            cfg.num_envs, cfg.seed0, cfg.random_target, cfg.is_discrete, cfg.rng_mode = n, 11, 1, 0, _lib.RNG_PHILOX
            h = _lib.Handle(cfg)
            obs0 = h.reset()
            out = h.rollout(T)
            ob = kuka_clib.body_trace(n)
            ora = kuka_clib.rollout(11 + np.arange(n), T, actions=None, is_discrete=False, random_target=True, rng_mode=kuka_clib.RNG_PHILOX, trace=False)
            kuka_clib.body_trace_off()
            assert np.array_equal(ora["actions"], out["actions"])
        elif mode == "joints_mt19937":
            n = 5
            cfg.num_envs, cfg.seed0, cfg.is_discrete, cfg.action_joints = n, 31, 0, 1
            actions = np.random.RandomState(5).uniform(-1, 1, size=(T, n, 7)).astype(np.float32)
            h = _lib.Handle(cfg)
            assert h.cfg.rng_mode == _lib.RNG_MT19937
            obs0 = h.reset()
            out = h.rollout(T, actions=actions)
            ob = kuka_clib.body_trace(n)
            ora = kuka_clib.rollout(31 + np.arange(n), T, actions=actions, is_discrete=False, action_joints=True, trace=False)
            kuka_clib.body_trace_off()
        else:
            from oracle import gym_seeding
            n = 6
            rs = np.random.RandomState(8)
            actions = rs.randint(6, size=(T, n)).astype(np.int32)
            actions[rs.rand(T, n) < 0.4] = 4
            cfg.num_envs, cfg.rng_mode, cfg.auto_reset = n, _lib.RNG_HOST, 0
            rngs = [gym_seeding.np_random(900 + i)[0] for i in range(n)]

            def reset_draws(r):                               # 20 uniforms of the ten distractor candidates, then the 5 init actions
                out = [r.uniform(-1, 1) for _ in range(20)]
                for _ in range(5):
                    out += [r.rand(), float(r.randint(3))]
                return out
            h = _lib.Handle(cfg)
            obs0 = h.reset(host_rand=np.array([reset_draws(r) for r in rngs]))
            ob = kuka_clib.body_trace(n)
            ora = kuka_clib.rollout(900 + np.arange(n), T, actions=actions, trace=False)
            kuka_clib.body_trace_off()
            out = {"obs": [], "reward": [], "done": []}
            for t in range(T):
                noise = np.array([r.normal(0.0, scale=0.01) for r in rngs])
                o, r_, d = h.step(actions[t], host_noise=noise)
                if d.any():
                    rand = np.zeros((n, 30))
                    for i in np.nonzero(d)[0]:
                        rand[i] = reset_draws(rngs[i])
                    o = h.reset(mask=d, host_rand=rand, obs_out=o.copy())
                out["obs"].append(o); out["reward"].append(r_); out["done"].append(d)
            out = {k: np.array(v) for k, v in out.items()}
    finally:
        kuka_clib.set_variant(kuka_clib.VARIANT_BUTTON)
        kuka_clib.body_trace_off()
    check_planes(ora, obs0, out)
    bodies = h.get_state(_lib.F_KUKA_BODIES).T.reshape(n, 11, 6)
    on = ob[:, :, 6] > 0
    assert np.abs(bodies - ob[:, :, :6])[on].max() <= 1e-9
    h.close()


@pytest.mark.parametrize("budget", [1, 2, 6])
def test_contact_sweep_instantiations_by_row_budget(budget):
    """The register-resident contact sweeps are compiled per number of bank-B slots in use (cn_sweeps<NG>, hand-scheduled bank-A
    rows).  With the reference's geometry a press brings ONE gripper sphere to the button, rarely two, almost never three (oracle:
    283 / 15 / 0 of 38 400 env-steps), so NG = 1 and 2 are the instantiations that run; the model table's row budget caps the number:
    budget 1 drops the second contact, 2 and 6 keep it — each against the oracle with the same budget."""
    n, T = 192, 500
    rs = np.random.RandomState(70 + budget)
    actions = rs.randint(6, size=(T, n)).astype(np.int32)
    actions[rs.rand(T, n) < 0.35] = 4
    t = _lib.kuka_tree_default_model().copy()
    t[kuka_model.TREE_MAX_GENERIC_ROWS] = float(budget)
    try:
        h = make(n, seed0=29)
        h.set_kuka_tree_model(t)
        obs0 = h.reset()
        out = h.rollout(T, actions=actions)
        kuka_clib.set_tree_model(t)
        ora = kuka_clib.rollout(29 + np.arange(n), T, actions=actions, aux=True, trace=False)
        normals = ora["rows"][:, :, 0]
        assert min(budget, 2) <= normals.max() <= budget and (normals >= min(budget, 2)).sum() > 5
        check_planes(ora, obs0, out)
        q = np.concatenate([h.get_state(_lib.F_KUKA_Q).T, h.get_state(_lib.F_KUKA_GRIPPER_Q).T], axis=1)
        f = ora["final_state"]
        assert np.abs(q - np.concatenate([f[:, :7], f[:, 30:35]], axis=1)).max() <= TOL
        h.close()
    finally:
        kuka_clib.set_full(True)


def test_configuration_specialised_instantiation_equals_the_generic_one(tmp_path):
    """Round 5: a handle with the reference's default KukaButtonGymEnv configuration on Philox streams runs kuka_tree_rollout_k's SPEC = 1
    instantiation (that configuration folded in as compile-time constants).  SRLHIP_KUKA_SPEC=0 (read once per process) keeps the
    generic one: a child process produces the same rollout with it — reward / done / sampled actions bit for bit, observations and
    final joints to 1e-9 (the two instantiations are scheduled differently, the arithmetic is the same)."""
    import os
    import subprocess
    import sys
    n, T = 512, 1100
    path = str(tmp_path / "generic.npz")
    code = ("import sys, numpy as np; sys.path[:0] = {!r}; import torch; from srlhip import _lib; "
            "cfg = _lib.default_config(_lib.ENV_KUKA_BUTTON); cfg.num_envs, cfg.rng_mode, cfg.auto_reset = {}, _lib.RNG_PHILOX, 1; "
            "h = _lib.Handle(cfg); o0 = h.reset(); out = h.rollout({}); "
            "np.savez({!r}, obs0=o0, q=h.get_state(_lib.F_KUKA_Q), **out)").format(sys.path[:4], n, T, path)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SRLHIP_KUKA_SPEC="0"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    gen = np.load(path)
    h = make(n, rng_mode=_lib.RNG_PHILOX)
    obs0 = h.reset()
    out = h.rollout(T)
    assert np.array_equal(obs0, gen["obs0"])
    for k in ("reward", "done", "actions"):
        assert np.array_equal(out[k], gen[k]), k
    assert np.abs(out["obs"] - gen["obs"]).max() <= 1e-6 and np.abs(h.get_state(_lib.F_KUKA_Q) - gen["q"]).max() <= 1e-9
    assert out["done"].sum() >= n
    h.close()
