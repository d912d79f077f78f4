"""GPU: persistent stepping (srlhip_set_persistent, HipVecEnv(persistent=True)) — the per-step API of the drop-in
(rl_baselines/utils.py:213-229: SubprocVecEnv.step_async / step_wait) without a kernel launch per step: one launch of the rollout kernel
stays resident with every env's state in registers and takes its steps from the host through mapped memory.  Same kernel code, so the
bar is bit-exact equality with the launching path — observations, rewards, dones, infos, Monitor files — across parks and restarts
(other API calls in between, the idle timeout), and a measurable latency gain."""
import os
import time

import numpy as np
import pytest

from srlhip import _lib
from srlhip.vec_env import HipVecEnv

pytestmark = pytest.mark.gpu


def _strip_t(infos):
    return [{k: ({kk: vv for kk, vv in v.items() if kk != "t"} if k == "episode" else v) for k, v in d.items()} for d in infos]


@pytest.mark.parametrize("n, rng_mode", [(4096, "mt19937"), (1000, "philox"), (7, "mt19937")])
def test_persistent_env_is_the_launching_env_bit_for_bit(n, rng_mode, tmp_path):
    kw = {"srl_model": "ground_truth"}
    a = HipVecEnv("KukaButtonGymEnv-v0", n, seed=3, env_kwargs=kw, rng_mode=rng_mode, log_dir=str(tmp_path / "a"))
    b = HipVecEnv("KukaButtonGymEnv-v0", n, seed=3, env_kwargs=kw, rng_mode=rng_mode, log_dir=str(tmp_path / "b"), persistent=True, park_us=3000)
    assert b.persistent and not a.persistent
    assert np.array_equal(a.reset(), b.reset())
    rs = np.random.RandomState(0)
    ended = 0
    for t in range(1250):
        act = rs.randint(6, size=n)
        x, y = a.step(act), b.step(act)
        assert np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1]) and np.array_equal(x[2], y[2]), t
        assert _strip_t(x[3]) == _strip_t(y[3]), t
        ended += int(x[2].sum())
        if t % 200 == 50:                  # another entry point parks the resident kernel; the next step restarts it
            assert np.array_equal(a._h.get_state(_lib.F_KUKA_Q), b._h.get_state(_lib.F_KUKA_Q))
        if t == 333:
            time.sleep(0.02)               # the idle timeout parks it too
        if t == 700:
            ia, ib = a.get_images(), b.get_images()
            assert all(np.array_equal(p, q) for p, q in zip(ia[:8], ib[:8]))
    assert ended >= n
    for u, v in zip(a.episode_returns(), b.episode_returns()):
        assert np.array_equal(u, v)
    a.close(); b.close()
    for i in range(n):
        with open(str(tmp_path / "a" / "{}.monitor.csv".format(i))) as f, open(str(tmp_path / "b" / "{}.monitor.csv".format(i))) as g:
            ra = [row.split(",")[:2] for row in f.read().splitlines()[2:]]
            rb = [row.split(",")[:2] for row in g.read().splitlines()[2:]]
        assert ra == rb, i


def test_persistent_is_refused_where_it_cannot_work():
    for kind, tweak in ((_lib.ENV_MOBILE, {"io_device": 1}), (_lib.ENV_KUKA_BUTTON, {"io_device": 1}), (_lib.ENV_KUKA_BUTTON, {"num_envs": 8192}),
                        (_lib.ENV_KUKA_RAND, {})):
        cfg = _lib.default_config(kind)
        cfg.num_envs, cfg.rng_mode = 64, _lib.RNG_PHILOX
        for k, v in tweak.items():
            setattr(cfg, k, v)
        h = _lib.Handle(cfg)
        with pytest.raises(_lib.SrlHipError):
            h.set_persistent(True)
        h.close()
    with pytest.raises(_lib.SrlHipError):
        HipVecEnv("KukaRandButtonGymEnv-v0", 16, env_kwargs={"srl_model": "ground_truth"}, persistent=True)
    env = HipVecEnv("MobileRobotGymEnv-v0", 16, env_kwargs={"srl_model": "ground_truth"})      # SRLHIP_PERSISTENT unset: plain launches
    assert not env.persistent
    env.close()


def test_persistent_step_latency():
    """steady state, 4096 envs, the reference's MT19937 streams: median step of the launching path against the resident kernel"""
    res = {}
    for name, persistent in (("launch per step", False), ("resident kernel", True)):
        env = HipVecEnv("KukaButtonGymEnv-v0", 4096, seed=0, env_kwargs={"srl_model": "ground_truth"}, persistent=persistent)
        env.reset()
        acts = np.random.RandomState(0).randint(6, size=(1500, 4096))
        for t in range(1100):
            env.step(acts[t])
        ts = []
        for t in range(1100, 1500):
            t0 = time.perf_counter(); env.step(acts[t]); ts.append(time.perf_counter() - t0)
        res[name] = float(np.median(ts)) * 1e6
        env.close()
    print("HipVecEnv.step, 4096 envs, steady state: " + ", ".join("{} {:.1f} us".format(k, v) for k, v in res.items()))
    # (measured: 84 vs 94 us — profiles/r06_vecenv_latency.txt; the assertion only guards against a pathological resident path, so that a
    #  noisy box cannot fail the suite on a timing)
    assert res["resident kernel"] < 1.25 * res["launch per step"]


def test_persistent_shards_on_one_device_share_its_residency():
    """two shards of one process on ONE device: both resident kernels fit (2 x 2048 envs) and the sharded env is the single one bit for
    bit; a second full-device handle is refused instead of waiting for room that never comes"""
    kw = {"srl_model": "ground_truth"}
    a = HipVecEnv("KukaButtonGymEnv-v0", 4096, seed=5, env_kwargs=kw)
    b = HipVecEnv("KukaButtonGymEnv-v0", 4096, seed=5, env_kwargs=kw, device_ids=[0, 0], persistent=True)
    assert b.persistent
    assert np.array_equal(a.reset(), b.reset())
    rs = np.random.RandomState(1)
    for t in range(300):
        act = rs.randint(6, size=4096)
        x, y = a.step(act), b.step(act)
        assert np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1]) and np.array_equal(x[2], y[2]), t
    with pytest.raises(_lib.SrlHipError):          # b's two shards hold the whole device
        HipVecEnv("KukaButtonGymEnv-v0", 64, seed=5, env_kwargs=kw, persistent=True)
    b.close()
    c = HipVecEnv("KukaButtonGymEnv-v0", 64, seed=5, env_kwargs=kw, persistent=True)     # released with the handles
    assert c.persistent
    c.close(); a.close()


@pytest.mark.parametrize("env_id, kw, rng_mode", [
    ("KukaButtonGymEnv-v0", {"srl_model": "ground_truth", "random_target": True}, "mt19937"),
    ("KukaButtonGymEnv-v0", {"srl_model": "ground_truth", "is_discrete": False, "shape_reward": True}, "philox"),
    ("KukaButtonGymEnv-v0", {"srl_model": "joints_position", "is_discrete": False, "action_joints": True}, "mt19937"),
    ("KukaButtonGymEnv-v0", {"srl_model": "joints", "action_repeat": 2}, "philox"),
    ("KukaMovingButtonGymEnv-v0", {"srl_model": "ground_truth"}, "mt19937"),
    ("Kuka2ButtonGymEnv-v0", {"srl_model": "ground_truth"}, "mt19937"),
    ("Kuka2ButtonGymEnv-v0", {"srl_model": "joints_position"}, "philox"),
])
def test_persistent_generic_instantiations(env_id, kw, rng_mode):
    """the one-button configurations the configuration-specialised kernel does not cover run the generic persistent instantiations:
    random targets, continuous Cartesian and joint-space actions, the joints observation modes, action repeat, the moving button, the
    two-button env"""
    n = 520
    a = HipVecEnv(env_id, n, seed=11, env_kwargs=kw, rng_mode=rng_mode)
    b = HipVecEnv(env_id, n, seed=11, env_kwargs=kw, rng_mode=rng_mode, persistent=True)
    assert b.persistent
    assert np.array_equal(a.reset(), b.reset())
    rs = np.random.RandomState(2)
    ended = 0
    for t in range(700):
        act = rs.randint(a.action_space.n, size=n) if kw.get("is_discrete", True) else rs.uniform(-1, 1, size=(n,) + a.action_space.shape).astype(np.float32)
        x, y = a.step(act), b.step(act)
        assert np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1]) and np.array_equal(x[2], y[2]), t
        ended += int(x[2].sum())
        if t == 400:
            assert np.array_equal(a._h.get_state(_lib.F_KUKA_Q), b._h.get_state(_lib.F_KUKA_Q))
    assert ended > 0 or kw.get("action_joints")
    a.close(); b.close()


def test_persistent_staged_output_path(monkeypatch):
    """the output path that is valid on ANY workgroup placement (staging copy in device memory + one copier per eighth of the grid): the
    default takes it only when an eighth of the grid does not sit on one XCD, SRLHIP_PERSIST_STAGED=1 forces it"""
    monkeypatch.setenv("SRLHIP_PERSIST_STAGED", "1")
    n, kw = 1000, {"srl_model": "ground_truth"}
    a = HipVecEnv("KukaButtonGymEnv-v0", n, seed=9, env_kwargs=kw)
    b = HipVecEnv("KukaButtonGymEnv-v0", n, seed=9, env_kwargs=kw, persistent=True)
    assert np.array_equal(a.reset(), b.reset())
    rs = np.random.RandomState(4)
    for t in range(600):
        act = rs.randint(6, size=n)
        x, y = a.step(act), b.step(act)
        assert np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1]) and np.array_equal(x[2], y[2]), t
        if t % 150 == 70:
            assert np.array_equal(a._h.get_state(_lib.F_KUKA_Q), b._h.get_state(_lib.F_KUKA_Q))      # park + restart
    a.close(); b.close()


@pytest.mark.parametrize("n", [4096, 1030, 5])
def test_single_step_launches_with_the_early_completion_signal_equal_the_fused_rollout(n):
    """the default per-step path does not wait for the kernel's end (the step's outputs are reported per XCD after one L2 write-back):
    1200 single steps — other entry points in between — against ONE fused 1200-step rollout of a twin handle on the same actions
    (no signal on that path: its planes are device memory), bit for bit, and Monitor's records with them"""
    T = 1200
    cfg = _lib.default_config(_lib.ENV_KUKA_BUTTON)
    cfg.num_envs, cfg.seed0, cfg.rng_mode, cfg.info_bits = n, 21, _lib.RNG_MT19937, 1
    a, b = _lib.Handle(cfg), _lib.Handle(cfg)
    assert np.array_equal(a.reset(), b.reset())
    acts = np.random.RandomState(3).randint(6, size=(T, n)).astype(np.int32)
    ref = b.rollout(T, actions=acts)
    out = (a.new_obs(), np.zeros(n, np.float32), np.zeros(n, np.uint8))
    ret, length = a.episode_records()
    ends = 0
    for t in range(T):
        if t % 2:
            a.step(acts[t], out=out)
        else:
            a.step_async(acts[t]); a.step_wait(out=out)
        assert np.array_equal(out[0], ref["obs"][t]) and np.array_equal(out[1], ref["reward"][t]) and np.array_equal(out[2], ref["done"][t]), t
        fin = np.flatnonzero(out[2] & 1)
        if fin.size:                              # Monitor's record of an episode that ended in this step is there when the step returns
            ends += fin.size
            assert (length[fin] > 0).all() and np.isfinite(ret[fin]).all()
        if t % 211 == 100:
            a.get_state(_lib.F_KUKA_Q)            # a synchronising entry point between two signalled steps
    assert ends >= n
    assert np.array_equal(a.get_state(_lib.F_KUKA_Q), b.get_state(_lib.F_KUKA_Q))
    sa, sb = a.episode_stats(), b.episode_stats()
    assert all(np.array_equal(x, y) for x, y in zip(sa, sb))
    a.close(); b.close()


def test_random_call_sequences_on_the_three_per_step_paths():
    """fuzz: random batch sizes, env variants, RNG modes and random interleavings of step / step_async + step_wait / masked reset /
    re-seed / state reads on three twin handles — single-step launches with the early completion signal, persistent stepping, and
    srlhip_rollout(T = 1) (no signal, no residency: staged copies and a stream synchronisation) — must agree bit for bit"""
    rs = np.random.RandomState(12345)
    kinds = [(_lib.ENV_KUKA_BUTTON, {}), (_lib.ENV_KUKA_BUTTON, {"random_target": 1}), (_lib.ENV_KUKA_MOVING, {"shape_reward": 1}),
             (_lib.ENV_KUKA_2BUTTON, {}), (_lib.ENV_KUKA_BUTTON, {"obs_mode": _lib.OBS_JOINTS})]
    for trial in range(30):
        kind, tweak = kinds[trial % len(kinds)]
        n = int(rs.choice([1, 3, 64, 257, 1000, 2048, 4096]))
        cfg = _lib.default_config(kind)
        cfg.num_envs, cfg.seed0, cfg.info_bits = n, 1000 * trial, 1
        cfg.rng_mode = _lib.RNG_PHILOX if trial % 2 else _lib.RNG_MT19937
        for k, v in tweak.items():
            setattr(cfg, k, v)
        hs = [_lib.Handle(cfg) for _ in range(3)]
        hs[1].set_persistent(True, 500)
        na = hs[0].num_actions
        obs = [h.reset() for h in hs]
        assert np.array_equal(obs[0], obs[1]) and np.array_equal(obs[0], obs[2])
        outs = [(h.new_obs(), np.zeros(n, np.float32), np.zeros(n, np.uint8)) for h in hs[:2]]
        for t in range(200):
            op = rs.rand()
            if op < 0.08:
                mask = (rs.rand(n) < 0.3).astype(np.uint8)
                r = [h.reset(mask=mask) for h in hs]
                assert np.array_equal(r[0], r[1]) and np.array_equal(r[0], r[2]), (trial, t)
                continue
            if op < 0.12:
                seeds = rs.randint(0, 2 ** 31, size=n).astype(np.int64)
                for h in hs:
                    h.seed(seeds)
                    h.reset()
                continue
            if op < 0.2:
                q = [h.get_state(_lib.F_KUKA_Q) for h in hs]
                assert np.array_equal(q[0], q[1]) and np.array_equal(q[0], q[2]), (trial, t)
                continue
            a = rs.randint(-1, na, size=n).astype(np.int32)            # (-1: the reference's `None` action)
            for h, o in zip(hs[:2], outs):
                if rs.rand() < 0.5:
                    h.step(a, out=o)
                else:
                    h.step_async(a); h.step_wait(out=o)
            ref = hs[2].rollout(1, actions=a[None])
            for o in outs:
                assert np.array_equal(o[0], ref["obs"][0]) and np.array_equal(o[1], ref["reward"][0]) and np.array_equal(o[2], ref["done"][0]), (trial, t, n)
        st = [h.episode_stats() for h in hs]
        assert all(np.array_equal(x, y) for x, y in zip(st[0], st[1])) and all(np.array_equal(x, y) for x, y in zip(st[0], st[2]))
        for h in hs:
            h.close()


@pytest.mark.parametrize("env_id, kw, rng_mode, n", [
    ("MobileRobotGymEnv-v0", {"srl_model": "ground_truth"}, "mt19937", 4096),
    ("MobileRobotGymEnv-v0", {"srl_model": "ground_truth", "is_discrete": False, "random_target": True}, "philox", 1000),
    ("MobileRobot1DGymEnv-v0", {"srl_model": "ground_truth"}, "mt19937", 7),
    ("MobileRobot2TargetGymEnv-v0", {"srl_model": "ground_truth", "shape_reward": True}, "philox", 300),
    ("MobileRobotLineTargetGymEnv-v0", {"srl_model": "ground_truth", "random_target": True}, "mt19937", 2049),
])
def test_persistent_mobile_robot_family(env_id, kw, rng_mode, n, tmp_path, monkeypatch):
    """the MobileRobot family under persistent stepping (mobile_persist_k): observations, rewards, dones, infos and Monitor files of the
    launching path bit for bit, across parks; the written-through output form (SRLHIP_PERSIST_STAGED=1) once as well"""
    if n == 300:
        monkeypatch.setenv("SRLHIP_PERSIST_STAGED", "1")
    a = HipVecEnv(env_id, n, seed=4, env_kwargs=kw, rng_mode=rng_mode, log_dir=str(tmp_path / "a"))
    b = HipVecEnv(env_id, n, seed=4, env_kwargs=kw, rng_mode=rng_mode, log_dir=str(tmp_path / "b"), persistent=True)
    assert b.persistent
    assert np.array_equal(a.reset(), b.reset())
    rs = np.random.RandomState(8)
    ended = 0
    for t in range(600):
        act = rs.randint(a.action_space.n, size=n) if kw.get("is_discrete", True) else rs.uniform(-1, 1, size=(n,) + a.action_space.shape).astype(np.float32)
        x, y = a.step(act), b.step(act)
        assert np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1]) and np.array_equal(x[2], y[2]), t
        assert _strip_t(x[3]) == _strip_t(y[3]), t
        ended += int(x[2].sum())
        if t % 170 == 60:
            assert np.array_equal(a._h.get_state(_lib.F_POS_X), b._h.get_state(_lib.F_POS_X))      # park + restart
    assert ended >= 2 * n
    a.close(); b.close()
    for i in (0, n - 1):
        with open(str(tmp_path / "a" / "{}.monitor.csv".format(i))) as f, open(str(tmp_path / "b" / "{}.monitor.csv".format(i))) as g:
            assert [r.split(",")[:2] for r in f.read().splitlines()[2:]] == [r.split(",")[:2] for r in g.read().splitlines()[2:]]
