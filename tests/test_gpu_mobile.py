"""GPU parity, MobileRobot family: HIP path (through the C-ABI) vs the CPU
oracle on the same seeds/actions.  Integer/flag outputs and the f64 internal
state are compared BIT-EXACT; observations bit-exact after the f32 cast."""
import os

import numpy as np
import pytest

from oracle import clib, mobile_oracle
from srlhip import _lib

pytestmark = pytest.mark.gpu

N_ACT = {0: 4, 1: 2, 2: 4, 3: 4}


def make(kind, n, rng_mode, auto_reset=1, seed0=0, first=0, **kw):
    cfg = _lib.default_config(kind)
    cfg.num_envs, cfg.rng_mode, cfg.auto_reset, cfg.seed0, cfg.first_env_id = n, rng_mode, auto_reset, seed0, first
    for k, v in kw.items():
        setattr(cfg, k, v)
    return _lib.Handle(cfg)


def assert_same(oracle, obs0, obs, rew, done, tag=""):
    assert np.array_equal(oracle["obs0"], obs0), tag + " obs0"
    assert np.array_equal(oracle["obs"], obs), tag + " obs"
    assert np.array_equal(oracle["reward"], rew), tag + " reward"
    assert np.array_equal(oracle["done"], done), tag + " done"


@pytest.mark.parametrize("kind", [0, 1, 2, 3])
@pytest.mark.parametrize("random_target", [0, 1])
def test_mt19937_step_by_step_matches_oracle(kind, random_target):
    """Reference-exact mode: env i seeded seed0+i like makeEnv (environments/utils.py:52);
    np_random lives on the device.  2+ episodes, per-step launches."""
    n, T, seed0 = 256, 2 * 251 + 9, 11
    rs = np.random.RandomState(1234)
    actions = rs.randint(N_ACT[kind], size=(T, n)).astype(np.int32)
    h = make(kind, n, _lib.RNG_MT19937, seed0=seed0, random_target=random_target)
    obs0 = h.reset()
    obs, rew, done = [], [], []
    for t in range(T):
        o, r, d = h.step(actions[t])
        obs.append(o.copy()); rew.append(r.copy()); done.append(d.copy())
    ora = clib.mobile_rollout(kind, seed0 + np.arange(n), T, actions=actions, random_target=bool(random_target))
    assert_same(ora, obs0, np.array(obs), np.array(rew), np.array(done), "kind%d" % kind)
    # internal float64 state, bit-exact
    fs = ora["final_state"]
    assert np.array_equal(h.get_state(_lib.F_POS_X), fs[:, 0])
    assert np.array_equal(h.get_state(_lib.F_POS_Y), fs[:, 1])
    assert np.array_equal(h.get_state(_lib.F_STEP_COUNT), fs[:, 6].astype(np.int32))
    ret, length, fin = h.episode_stats()
    assert np.array_equal(ret, ora["ep_stats"][:, 0])
    assert np.array_equal(length, ora["ep_stats"][:, 1].astype(np.int32))
    assert np.array_equal(fin, ora["ep_stats"][:, 2].astype(np.int32))
    assert fin.min() == 2
    h.close()


@pytest.mark.parametrize("kind", [0, 1, 2, 3])
def test_fused_rollout_4096_envs_matches_oracle(kind):
    """BASELINE config 2 size: 4096 envs, fused T-step kernel, host actions, MT19937 streams."""
    n, T = 4096, 2 * 251 + 3
    rs = np.random.RandomState(99)
    actions = rs.randint(N_ACT[kind], size=(T, n)).astype(np.int32)
    h = make(kind, n, _lib.RNG_MT19937, seed0=0, shape_reward=1 if kind == 0 else 0)
    obs0 = h.reset()
    out = h.rollout(T, actions=actions)
    ora = clib.mobile_rollout(kind, np.arange(n), T, actions=actions, shape_reward=(kind == 0))
    assert_same(ora, obs0, out["obs"], out["reward"], out["done"])
    if kind == 0:   # shaped reward: uncast float64 of the last step
        assert np.array_equal(h.get_state(_lib.F_LAST_REWARD), ora["reward64"][-1])
    h.close()


@pytest.mark.parametrize("discrete", [1, 0])
def test_philox_random_agent_rollout_matches_oracle(discrete):
    """Throughput mode: Philox env stream + device-sampled random-agent actions,
    replayed bit-exactly by the oracle's Philox restatement."""
    n, T = 4096, 600
    h = make(0, n, _lib.RNG_PHILOX, seed0=5, is_discrete=discrete, random_target=1)
    obs0 = h.reset()
    out = h.rollout(T)
    ora = clib.mobile_rollout(0, 5 + np.arange(n), T, actions=None, is_discrete=bool(discrete), random_target=True,
                              rng_mode=clib.RNG_PHILOX)
    assert np.array_equal(ora["actions"], out["actions"])
    assert_same(ora, obs0, out["obs"], out["reward"], out["done"])
    # chunked rollouts continue the same streams
    h2 = make(0, n, _lib.RNG_PHILOX, seed0=5, is_discrete=discrete, random_target=1)
    h2.reset()
    a = h2.rollout(250)
    b = h2.rollout(T - 250)
    assert np.array_equal(np.concatenate([a["obs"], b["obs"]]), out["obs"])
    h.close(); h2.close()


@pytest.mark.parametrize("kind,random_target,discrete", [(0, 1, 1), (1, 1, 1), (2, 1, 1), (3, 0, 1), (0, 0, 0)])
def test_episode_parallel_rollout_equals_step_by_step(kind, random_target, discrete):
    """The Philox rollout runs the segments of an env's rollout (rest of the running episode, then whole 251-step
    episodes) on separate lanes (mobile_rollout_ep_k); everything it leaves behind — output planes, env state, RNG
    counters, episode statistics — has to equal T single-step launches of the sequential kernel.  Episode clocks are
    desynchronised first (masked resets, odd chunk lengths) so that wavefronts mix different segment boundaries."""
    n, T = 333, 3 * 251 + 77
    rs = np.random.RandomState(kind + 10)
    if discrete:
        acts = rs.randint(N_ACT[kind], size=(T + 200, n)).astype(np.int32)
    else:
        acts = rs.uniform(-1.2, 1.2, size=(T + 200, n, 2)).astype(np.float32)
    hs = [make(kind, n, _lib.RNG_PHILOX, seed0=21, random_target=random_target, is_discrete=discrete, shape_reward=1 if kind == 2 else 0) for _ in range(2)]
    for h in hs:
        h.reset()
        h.rollout(70, actions=acts[:70])                               # T >= 32: already the episode-parallel kernel
        mask = (np.arange(n) % 3 == 0).astype(np.uint8)
        h.reset(mask=mask)                                             # a third of the envs restart their episode clock
        for t in range(70, 113):
            h.step(acts[t])
        h.reset(mask=(np.arange(n) % 7 == 1).astype(np.uint8))
    assert len(np.unique(hs[0].get_state(_lib.F_STEP_COUNT))) >= 3
    a, b = hs
    out = a.rollout(T, actions=acts[200:200 + T])
    obs, rew, done = [], [], []
    for t in range(T):
        o, r, d = b.step(acts[200 + t])
        obs.append(o.copy()); rew.append(r.copy()); done.append(d.copy())
    assert np.array_equal(out["obs"], np.stack(obs)) and np.array_equal(out["reward"], np.stack(rew))
    assert np.array_equal(out["done"], np.stack(done)) and out["done"].sum() >= 3 * n
    for f in (_lib.F_POS_X, _lib.F_POS_Y, _lib.F_TARGET_X, _lib.F_TARGET_Y, _lib.F_TARGET2_X, _lib.F_TARGET2_Y,
              _lib.F_STEP_COUNT, _lib.F_CUR_TARGET, _lib.F_LAST_REWARD, _lib.F_EP_RETURN, _lib.F_EP_LENGTH):
        assert np.array_equal(a.get_state(f), b.get_state(f)), f
    for x, y in zip(a.episode_stats(), b.episode_stats()):
        assert np.array_equal(x, y)
    # and the streams continue identically afterwards (same RNG counters): one more episode-parallel rollout on both
    more = acts[:60]
    assert np.array_equal(a.rollout(60, actions=more)["obs"], b.rollout(60, actions=more)["obs"])
    # sampled actions: the action-stream counter advances by T as well
    if discrete:
        x, y = a.rollout(300), b.rollout(300)
        assert np.array_equal(x["actions"], y["actions"]) and np.array_equal(x["obs"], y["obs"])
    a.close(); b.close()


def test_bench_configuration_rollout_matches_oracle():
    """Exactly what bench.py --workload mobile times (BASELINE config 2): 4096 envs, one 2048-step device-sampled
    random-agent rollout after a warm-up rollout — every observation / reward / done of the 8.4M env-steps against the
    C oracle's Philox restatement, plus the episode statistics the Monitor equivalent reports."""
    n, T = 4096, 2048
    h = make(0, n, _lib.RNG_PHILOX, seed0=0)
    obs0 = h.reset()
    warm = h.rollout(256)
    out = h.rollout(T)
    ora = clib.mobile_rollout(0, np.arange(n), 256 + T, actions=None, rng_mode=clib.RNG_PHILOX)
    assert np.array_equal(ora["obs0"], obs0)
    for k in ("actions", "obs", "reward", "done"):
        assert np.array_equal(ora[k][:256], warm[k]), k
        assert np.array_equal(ora[k][256:], out[k]), k
    ret, length, fin = h.episode_stats()
    assert (fin == (256 + T) // 251).all() and (length == 251).all()
    done_rows = np.nonzero(ora["done"][:, 0])[0]
    last = slice(done_rows[-2] + 1, done_rows[-1] + 1)
    assert np.array_equal(ret, ora["reward"][last].astype(np.float64).sum(axis=0))
    h.close()


def test_host_rng_mode_and_manual_reset():
    """RNG_HOST harness: every draw supplied by the caller (numpy RandomState per env),
    no auto-reset: finished envs are reset with srlhip_reset(mask) like a VecEnv worker."""
    from oracle import gym_seeding
    n, T = 64, 300
    rngs = [gym_seeding.np_random(100 + i)[0] for i in range(n)]
    envs = []
    for i in range(n):
        e = mobile_oracle.MobileOracleEnv(mobile_oracle.MOBILE, random_target=True)
        e.seed(100 + i)
        envs.append(e)
    h = make(0, n, _lib.RNG_HOST, auto_reset=0, random_target=1)

    def draw_reset(rng):
        return [rng.uniform(-4 / 3, 4 / 3), rng.uniform(-4 / 3, 4 / 3), rng.uniform(0.4, 3.6), rng.uniform(0.4, 3.6)]

    obs = h.reset(host_rand=np.array([draw_reset(r) for r in rngs]))
    ref = np.array([e.reset() for e in envs], dtype=np.float32)
    assert np.array_equal(obs, ref)
    arng = np.random.RandomState(3)
    for t in range(T):
        a = arng.randint(4, size=n).astype(np.int32)
        noise = np.array([r.normal(0.0, scale=0.0) for r in rngs])
        o, r_, d = h.step(a, host_noise=noise)
        exp = [e.step(int(a[i])) for i, e in enumerate(envs)]
        assert np.array_equal(o, np.array([x[0] for x in exp], dtype=np.float32))
        assert np.array_equal(r_, np.array([x[1] for x in exp], dtype=np.float32))
        assert np.array_equal(d.astype(bool), np.array([x[2] for x in exp]))
        if d.any():
            rand = np.zeros((n, 4))
            for i in np.nonzero(d)[0]:
                rand[i] = draw_reset(rngs[i])
            o2 = h.reset(mask=d, host_rand=rand, obs_out=o.copy())
            for i in np.nonzero(d)[0]:
                assert np.array_equal(o2[i], envs[i].reset().astype(np.float32))
            assert np.array_equal(o2[d == 0], o[d == 0])
    h.close()


def test_sharding_invariance_and_reseed():
    """Env streams depend on the GLOBAL env id only: 2 shards == 1 handle;
    srlhip_seed(mask) reseeds like dataset_generator.py:82-86."""
    n, T = 512, 300
    rs = np.random.RandomState(5)
    actions = rs.randint(4, size=(T, n)).astype(np.int32)
    full = make(0, n, _lib.RNG_MT19937, seed0=7, random_target=1)
    o_full = full.reset()
    r_full = full.rollout(T, actions=actions)
    parts = []
    for g in range(2):
        hs = make(0, n // 2, _lib.RNG_MT19937, seed0=7, first=g * n // 2, random_target=1)
        o = hs.reset()
        r = hs.rollout(T, actions=np.ascontiguousarray(actions[:, g * n // 2:(g + 1) * n // 2]))
        parts.append((o, r))
        hs.close()
    assert np.array_equal(np.concatenate([p[0] for p in parts]), o_full)
    for k in ("obs", "reward", "done"):
        assert np.array_equal(np.concatenate([p[1][k] for p in parts], axis=1), r_full[k])
    # reseed half of the envs with new seeds and reset only those
    mask = (np.arange(n) % 2).astype(np.uint8)
    full.seed(1000 + np.arange(n), mask=mask)
    o2 = full.reset(mask=mask, obs_out=np.zeros((n, 2), np.float32))
    ora = clib.mobile_rollout(0, 1000 + np.arange(n), 1, actions=np.zeros((1, n), np.int32), random_target=True)
    assert np.array_equal(o2[mask == 1], ora["obs0"][mask == 1])
    full.close()


def test_device_io_mode_with_torch_buffers():
    """io_device=1: torch owns the I/O buffers (plumbing only), zero host copies."""
    import torch
    n, T = 4096, 64
    h = make(0, n, _lib.RNG_PHILOX, seed0=1, io_device=1)
    dev = torch.device("cuda:0")
    obs0 = torch.zeros((n, 2), dtype=torch.float32, device=dev)
    h.reset(obs_out=obs0.data_ptr())
    obs = torch.zeros((T, n, 2), dtype=torch.float32, device=dev)
    rew = torch.zeros((T, n), dtype=torch.float32, device=dev)
    done = torch.zeros((T, n), dtype=torch.uint8, device=dev)
    act = torch.zeros((T, n), dtype=torch.int32, device=dev)
    h.rollout(T, out=(obs.data_ptr(), rew.data_ptr(), done.data_ptr(), act.data_ptr()))
    h.sync()
    ora = clib.mobile_rollout(0, 1 + np.arange(n), T, actions=None, rng_mode=clib.RNG_PHILOX)
    assert np.array_equal(obs0.cpu().numpy(), ora["obs0"])
    assert np.array_equal(obs.cpu().numpy(), ora["obs"])
    assert np.array_equal(act.cpu().numpy(), ora["actions"])
    assert np.array_equal(rew.cpu().numpy(), ora["reward"])
    h.close()


@pytest.mark.parametrize("T", [200, 400, 2200])
def test_episode_parallel_rollout_beyond_resident_lanes(T):
    """2^20 envs: the episode-parallel rollout launches up to 10 M lanes, far more than can be resident at once, so
    later segment lanes of an env start after the lane that stores the env's final state has retired.  The kernel reads
    the per-env state from a snapshot taken right before the launch; results have to equal the sequential kernel
    (chunks shorter than 32 steps) bit for bit: reward / done planes, state, RNG counters, episode statistics."""
    import torch
    n, chunk = 1 << 20, 25
    dev = torch.device("cuda", 0)
    hs = []
    for _ in range(2):
        cfg = _lib.default_config(_lib.ENV_MOBILE)
        cfg.num_envs, cfg.seed0, cfg.rng_mode, cfg.auto_reset, cfg.io_device = n, 5, _lib.RNG_PHILOX, 1, 1
        hs.append(_lib.Handle(cfg))
    a, b = hs
    planes = []
    for h in hs:
        h.reset(obs_out=0)
        h.rollout(70, out=(0, 0, 0, 0))                      # desynchronise from step 0 (counter 70)
        planes.append((torch.zeros((T, n), dtype=torch.float32, device=dev), torch.zeros((T, n), dtype=torch.uint8, device=dev)))
    a.rollout(T, out=(0, planes[0][0].data_ptr(), planes[0][1].data_ptr(), 0))
    t = 0
    while t < T:
        c = min(chunk, T - t)
        b.rollout(c, out=(0, planes[1][0][t].data_ptr(), planes[1][1][t].data_ptr(), 0))
        t += c
    a.sync(); b.sync()
    assert torch.equal(planes[0][0], planes[1][0]) and torch.equal(planes[0][1], planes[1][1])
    assert int(planes[0][1].sum()) >= n * ((T + 70) // 251)
    for f in (_lib.F_POS_X, _lib.F_POS_Y, _lib.F_STEP_COUNT, _lib.F_LAST_REWARD, _lib.F_EP_RETURN, _lib.F_EP_LENGTH,
              _lib.F_LAST_RETURN, _lib.F_LAST_LENGTH, _lib.F_N_FINISHED):
        assert np.array_equal(a.get_state(f), b.get_state(f)), f
    # the streams continue identically (RNG and action counters)
    a.rollout(40, out=(0, planes[0][0].data_ptr(), 0, 0)); b.rollout(20, out=(0, planes[1][0].data_ptr(), 0, 0))
    b.rollout(20, out=(0, planes[1][0][20].data_ptr(), 0, 0))
    a.sync(); b.sync()
    assert torch.equal(planes[0][0][:40], planes[1][0][:40])
    a.close(); b.close()


@pytest.mark.parametrize("kind,is_discrete", [(_lib.ENV_MOBILE, 1), (_lib.ENV_MOBILE, 0), (_lib.ENV_MOBILE_2TARGET, 1)])
def test_synthetic_agent_action_plane_drawn_ahead(kind, is_discrete):
    """Rollouts without caller actions: the NEXT rollout's action plane is drawn by spare workgroups of the current launch.  Same
    planes, observations, rewards and dones as with the standalone sampler (SRLHIP_NO_ACTION_PREFETCH=1), also across a change of
    T (the plane drawn ahead is dropped), a re-seed and a rollout with caller actions in between."""
    n = 512

    def run(prefetch):
        if prefetch:
            os.environ.pop("SRLHIP_NO_ACTION_PREFETCH", None)
        else:
            os.environ["SRLHIP_NO_ACTION_PREFETCH"] = "1"
        try:
            cfg = _lib.default_config(kind)
            cfg.num_envs, cfg.seed0, cfg.rng_mode, cfg.is_discrete, cfg.random_target = n, 3, _lib.RNG_PHILOX, is_discrete, 1
            h = _lib.Handle(cfg)
            h.reset()
            outs = [h.rollout(T) for T in (300, 300, 300, 100, 300)]
            given = outs[0]["actions"][:64].copy()
            outs.append(h.rollout(64, actions=given))                  # caller actions: the stream counters do not move
            outs.append(h.rollout(300))
            h.seed(np.arange(n, dtype=np.int64) + 77)
            h.reset()
            outs.append(h.rollout(300))
            outs.append(h.rollout(300))
            h.close()
            return outs
        finally:
            os.environ.pop("SRLHIP_NO_ACTION_PREFETCH", None)
    a, b = run(True), run(False)
    for x, y in zip(a, b):
        for k in ("obs", "reward", "done", "actions"):
            if k in x and x[k] is not None:
                assert np.array_equal(x[k], y[k]), k
    assert not np.array_equal(a[0]["actions"], a[1]["actions"])         # consecutive planes continue the stream


@pytest.mark.parametrize("kind,shape_reward", [(0, 1), (2, 1), (3, 1), (1, 1), (0, 0)])
def test_episode_parallel_kernel_instantiations(kind, shape_reward):
    """mobile_rollout_ep_k is compiled per (kind, action type, shaped reward, which output planes exist): the shaped reward's
    float64 sqrt, the straight-line chunks of interior steps with all four planes (synthetic agent), with three (the caller's
    actions) and the per-step checked loop (a plane missing) — every combination against the oracle's Philox restatement
    (ragged episode clocks inside a wavefront: test_episode_parallel_rollout_equals_step_by_step)."""
    n, T = 700, 2 * 251 + 61
    rs = np.random.RandomState(40 + kind)
    acts = rs.randint(N_ACT[kind], size=(T + 90, n)).astype(np.int32)
    seeds = 9 + np.arange(n)
    okw = dict(shape_reward=bool(shape_reward), random_target=True, rng_mode=clib.RNG_PHILOX)
    # (a) caller's actions, all output planes: three stores per step
    h = make(kind, n, _lib.RNG_PHILOX, seed0=9, shape_reward=shape_reward, random_target=1)
    obs0 = h.reset()
    out = h.rollout(T, actions=acts[:T])
    ora = clib.mobile_rollout(kind, seeds, T, actions=acts[:T], **okw)
    assert_same(ora, obs0, out["obs"], out["reward"], out["done"], "given")
    if shape_reward:
        assert np.array_equal(h.get_state(_lib.F_LAST_REWARD), ora["reward64"][-1])
    # (b) the same continued with a plane missing (reward not wanted): the checked loop; the state it leaves must be the same
    more = h.rollout(90, actions=acts[T:], want=("obs", "done"))
    ora2 = clib.mobile_rollout(kind, seeds, T + 90, actions=acts, **okw)
    assert np.array_equal(more["obs"], ora2["obs"][T:]) and np.array_equal(more["done"], ora2["done"][T:])
    h.close()
    # (c) synthetic agent: four stores per step; the second rollout starts 77 steps into the episodes
    h = make(kind, n, _lib.RNG_PHILOX, seed0=9, shape_reward=shape_reward, random_target=1)
    h.reset()
    a = h.rollout(77)
    b = h.rollout(T)
    ora = clib.mobile_rollout(kind, seeds, 77 + T, actions=None, **okw)
    assert np.array_equal(np.concatenate([a["actions"], b["actions"]]), ora["actions"])
    assert np.array_equal(np.concatenate([a["obs"], b["obs"]]), ora["obs"])
    assert np.array_equal(np.concatenate([a["reward"], b["reward"]]), ora["reward"])
    assert np.array_equal(np.concatenate([a["done"], b["done"]]), ora["done"])
    h.close()


def test_chained_rollouts_with_steps_and_state_writes_in_between():
    """Round 5: back-to-back episode-parallel rollouts read the snapshot the previous launch's final-state lanes wrote (no copy kernel);
    per-step calls, a masked reset and a set_state in between fall back to the copy.  The whole sequence equals the same handle
    configuration stepped one launch per step."""
    n, seed0 = 300, 21
    rs = np.random.RandomState(3)
    plan = [("rollout", 300), ("rollout", 40), ("steps", 5), ("rollout", 700), ("rollout", 33), ("reset_mask", 0), ("rollout", 260),
            ("set_state", 0), ("rollout", 100), ("rollout", 100)]
    ha = make(0, n, _lib.RNG_PHILOX, seed0=seed0, random_target=1)
    hb = make(0, n, _lib.RNG_PHILOX, seed0=seed0, random_target=1)
    assert np.array_equal(ha.reset(), hb.reset())
    mask = (np.arange(n) % 3 == 0).astype(np.uint8)
    for what, T in plan:
        if what in ("rollout", "steps"):
            actions = rs.randint(4, size=(T, n)).astype(np.int32)
            ob, rw, dn = np.zeros((T, n, 2), np.float32), np.zeros((T, n), np.float32), np.zeros((T, n), np.uint8)
            for t in range(T):
                ob[t], rw[t], dn[t] = hb.step(actions[t])
            if what == "rollout":
                out = ha.rollout(T, actions=actions)
                got = (out["obs"], out["reward"], out["done"])
            else:
                got = (np.zeros_like(ob), np.zeros_like(rw), np.zeros_like(dn))
                for t in range(T):
                    got[0][t], got[1][t], got[2][t] = ha.step(actions[t])
            assert np.array_equal(got[0], ob) and np.array_equal(got[1], rw) and np.array_equal(got[2], dn), (what, T)
        elif what == "reset_mask":
            assert np.array_equal(ha.reset(mask=mask), hb.reset(mask=mask))
        else:
            x = ha.get_state(_lib.F_POS_X) * 0.5
            ha.set_state(_lib.F_POS_X, x); hb.set_state(_lib.F_POS_X, x)
        for f in (_lib.F_POS_X, _lib.F_POS_Y, _lib.F_STEP_COUNT, _lib.F_EP_RETURN, _lib.F_EP_LENGTH, _lib.F_LAST_RETURN, _lib.F_N_FINISHED):
            assert np.array_equal(ha.get_state(f), hb.get_state(f)), (what, T, f)
    ha.close(); hb.close()
