"""GPU: the node-wide drop-in — HipVecEnv(device_ids=[...]), one process, G handles (rl_baselines/utils.py:194-229 builds ONE VecEnv
for the single-process rl_baselines/train.py:172-333; environments/utils.py:52 seeds env `rank` with seed + rank).

The GPU box has one device, so the shards of these tests all live on device 0 (device_ids=[0, 0, 0, 0]) — the host-side logic
(global env ids, first_env_id seeding, plane / infos / Monitor assembly in global order, step_async launching every shard before
step_wait collects any) is the same code that drives eight devices.  Bar: a sharded env is the single-handle env BIT FOR BIT
(observations, rewards, dones, infos, Monitor files), and the shards are in flight at once."""
import ctypes
import os
import time

import numpy as np
import pytest

import kuka_scripts
from srlhip import _lib
from srlhip.vec_env import HipVecEnv, ShardedHipVecEnv, shard_bounds

pytestmark = pytest.mark.gpu


def _monitor_rows(log_dir, n):
    """{global env id: [(r, l)]} of the Monitor CSV files (the t column is wall-clock)"""
    out = {}
    for i in range(n):
        with open(os.path.join(log_dir, "{}.monitor.csv".format(i))) as f:
            lines = f.read().splitlines()
        assert lines[0].startswith("#") and lines[1] == "r,l,t"
        out[i] = [tuple(row.split(",")[:2]) for row in lines[2:]]
    return out


def _strip_t(infos):
    return [{k: ({kk: vv for kk, vv in v.items() if kk != "t"} if k == "episode" else v) for k, v in d.items()} for d in infos]


@pytest.mark.parametrize("env_id, n, device_ids, steps", [
    ("KukaButtonGymEnv-v0", 4096, [0, 0, 0, 0], 1100),
    ("MobileRobotGymEnv-v0", 4096, [0, 0, 0, 0], 300),
    ("KukaButtonGymEnv-v0", 1000, [0, 0, 0], 1050),              # ragged shards: 334 + 333 + 333
    ("MobileRobot2TargetGymEnv-v0", 37, [0] * 8, 300),           # shards of 5 and 4 envs (tail groups of the kernels)
])
def test_sharded_env_is_the_single_handle_env_bit_for_bit(env_id, n, device_ids, steps, tmp_path):
    kw = {"srl_model": "ground_truth"}
    one = HipVecEnv(env_id, n, seed=5, env_kwargs=kw, log_dir=str(tmp_path / "one"))
    many = ShardedHipVecEnv(env_id, n, device_ids, seed=5, env_kwargs=kw, log_dir=str(tmp_path / "many"))
    assert len(many._shards) == len(device_ids) and [(s.lo, s.hi) for s in many._shards] == shard_bounds(n, len(device_ids))
    assert [s.h.cfg.first_env_id for s in many._shards] == [s.lo for s in many._shards]
    assert np.array_equal(one.reset(), many.reset())
    rs = np.random.RandomState(1)
    ended = 0
    for t in range(steps):
        a = rs.randint(one.action_space.n, size=n)
        o1, r1, d1, i1 = one.step(a)
        o2, r2, d2, i2 = many.step(a)
        assert np.array_equal(o1, o2) and np.array_equal(r1, r2) and np.array_equal(d1, d2), t
        assert _strip_t(i1) == _strip_t(i2), t
        ended += int(d1.sum())
    assert ended >= n // 2                                        # episode records really crossed shard boundaries
    for a, b in zip(one.episode_returns(), many.episode_returns()):
        assert np.array_equal(a, b)
    one.close(); many.close()
    assert _monitor_rows(str(tmp_path / "one"), n) == _monitor_rows(str(tmp_path / "many"), n)


def test_sharded_rollout_and_images_match_the_single_handle():
    n, T = 512, 300
    kw = {"srl_model": "ground_truth", "img_shape": (64, 64)}
    one = HipVecEnv("KukaButtonGymEnv-v0", n, seed=2, env_kwargs=kw, rng_mode="philox", allow_early_resets=True)
    many = HipVecEnv("KukaButtonGymEnv-v0", n, seed=2, env_kwargs=kw, rng_mode="philox", device_ids=[0, 0, 0, 0], allow_early_resets=True)
    assert np.array_equal(one.reset(), many.reset())
    a, b = one.rollout(T), many.rollout(T)                 # device-side Philox agent: the action stream is keyed by the global env id
    for k in ("obs", "reward", "done", "actions"):
        assert np.array_equal(a[k], b[k]), k
    acts = np.random.RandomState(0).randint(6, size=(40, n)).astype(np.int32)
    a, b = one.rollout(40, actions=acts), many.rollout(40, actions=acts)
    for k in ("obs", "reward", "done"):
        assert np.array_equal(a[k], b[k]), k
    ia, ib = one.get_images(), many.get_images()
    assert len(ia) == len(ib) == n and all(np.array_equal(x, y) for x, y in zip(ia, ib))
    one.seed(77); many.seed(77)
    assert np.array_equal(one.reset(), many.reset())
    one.close(); many.close()


def test_raw_pixels_observations_sharded():
    n = 24
    kw = {"srl_model": "raw_pixels", "img_shape": (64, 64)}
    one = HipVecEnv("MobileRobotGymEnv-v0", n, seed=1, env_kwargs=kw)
    many = HipVecEnv("MobileRobotGymEnv-v0", n, seed=1, env_kwargs=kw, device_ids=[0, 0, 0])
    o1, o2 = one.reset(), many.reset()
    assert o1.dtype == np.uint8 and o1.shape == (n, 64, 64, 3) and np.array_equal(o1, o2) and o1.any()
    rs = np.random.RandomState(3)
    for t in range(20):
        a = rs.randint(4, size=n)
        x, y = one.step(a), many.step(a)
        assert all(np.array_equal(p, q) for p, q in zip(x[:3], y[:3]))
    one.close(); many.close()


def test_learned_srl_model_sharded_one_encoder_replica_per_shard(tmp_path):
    """srl_model = a learned encoder (rl_baselines/utils.py:213-216: one MultiprocessSRLModel server for all envs): frames stay on the
    device, every shard encodes its own on its own stream; states / reward / done / infos equal the single-handle env's."""
    import torch
    from state_representation.models import SRLNeuralNetwork
    torch.manual_seed(0)
    n = 96
    enc = SRLNeuralNetwork(5, cuda=True, img_shape=(64, 64), backend="hip")
    kw = {"srl_model": "autoencoder", "img_shape": (64, 64)}
    one = HipVecEnv("KukaButtonGymEnv-v0", n, seed=9, env_kwargs=kw, encoder=enc)
    many = HipVecEnv("KukaButtonGymEnv-v0", n, seed=9, env_kwargs=kw, encoder=enc, device_ids=[0, 0, 0], log_dir=str(tmp_path))
    twin = enc.replicate(torch.device("cuda", 0))                    # what a shard on another GPU would get
    imgs = np.random.RandomState(1).randint(0, 256, size=(7, 64, 64, 3)).astype(np.uint8)
    assert torch.equal(twin.getStates(imgs), enc.getStates(imgs))
    o1, o2 = one.reset(), many.reset()
    assert o1.shape == (n, 5) and o1.dtype == np.float32 and np.array_equal(o1, o2)
    rs = np.random.RandomState(4)
    for t in range(150):
        a = rs.randint(6, size=n)
        x, y = one.step(a), many.step(a)
        assert np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1]) and np.array_equal(x[2], y[2]), t
        assert _strip_t(x[3]) == _strip_t(y[3])
    ia, ib = one.get_images(), many.get_images()
    assert all(np.array_equal(p, q) for p, q in zip(ia, ib))
    one.close(); many.close()


def test_step_async_launches_and_the_shards_run_concurrently():
    """(a) step_async returns while the GPU is still stepping (the launch happens THERE: round 5's step_async only stored the actions);
    (b) two 2048-env shards on one device take about as long as ONE of them, not twice: both are in flight before either is
    collected (a 2048-env Kuka launch is 512 wavefronts — half the SIMDs of the chip — so two of them fit side by side)."""
    kw = {"srl_model": "ground_truth"}

    def median_step(env, reps=60):
        rs = np.random.RandomState(0)
        env.reset()
        acts = rs.randint(6, size=(reps + 10, env.num_envs))
        t_async, t_total = [], []
        for t in range(reps + 10):
            t0 = time.perf_counter()
            env.step_async(acts[t])
            t1 = time.perf_counter()
            env.step_wait()
            t2 = time.perf_counter()
            t_async.append(t1 - t0); t_total.append(t2 - t0)
        return float(np.median(t_async[10:])), float(np.median(t_total[10:]))

    single = HipVecEnv("KukaButtonGymEnv-v0", 2048, seed=0, env_kwargs=kw)
    a1, s1 = median_step(single)
    single.close()
    two = HipVecEnv("KukaButtonGymEnv-v0", 4096, seed=0, env_kwargs=kw, device_ids=[0, 0])
    a2, s2 = median_step(two)
    two.close()
    print("one 2048-env shard: step {:.1f} us (step_async {:.1f} us); two shards on one device: step {:.1f} us (step_async {:.1f} us)".format(
        s1 * 1e6, a1 * 1e6, s2 * 1e6, a2 * 1e6))
    assert a1 < 0.6 * s1, "step_async should return long before the step is done"
    assert s2 <= 1.3 * s1 + 15e-6, "two shards in flight at once must not cost two steps"      # (+15 us: the second shard's launch + collect calls)


def test_abi_step_async_wait_contract():
    cfg = _lib.default_config(_lib.ENV_MOBILE)
    cfg.num_envs, cfg.rng_mode = 64, _lib.RNG_PHILOX
    h = _lib.Handle(cfg)
    twin = _lib.Handle(cfg)
    h.reset(); twin.reset()
    lib = _lib.load()
    a = np.random.RandomState(0).randint(4, size=(5, 64)).astype(np.int32)
    assert lib.srlhip_step_wait(h._h, None, None, None) == -22 and b"no srlhip_step_async" in lib.srlhip_last_error(h._h)
    for t in range(5):
        assert not h.step_pending()
        h.step_async(a[t])
        assert h.step_pending()
        assert lib.srlhip_step_async(h._h, a[t].ctypes.data_as(ctypes.c_void_p), None) == -22           # one step in flight per handle
        with pytest.raises(_lib.SrlHipError):
            h.step(a[t])
        got = h.step_wait()
        want = twin.step(a[t])
        assert all(np.array_equal(x, y) for x, y in zip(got, want))
    # device-pointer handles: srlhip_step already only enqueues
    cfg.io_device = 1
    d = _lib.Handle(cfg)
    assert lib.srlhip_step_async(d._h, a[0].ctypes.data_as(ctypes.c_void_p), None) == -22
    d.close(); h.close(); twin.close()


def test_infos_carry_the_ik_conditioning_flag():
    """behind SRLHIP_F_KUKA_IK_CROSSED the 1e-4 / bit-exact bar is not claimed — a VecEnv caller must be able to see
    it.  Saturating scripts (tests/kuka_scripts.py: +x held walks the elbow through its singularity): infos[i]["ik_crossed"] is set on
    exactly the env-steps the handle's flag plane counts, dones stay 0 / 1, and a random agent never sees the key."""
    names, ss, actions = kuka_scripts.batch(kuka_scripts.discrete_scripts(), (7, 8))
    n, T = len(ss), 700
    env = HipVecEnv("KukaButtonGymEnv-v0", n, seed=0, env_kwargs={"srl_model": "ground_truth"}, device_ids=[0, 0])
    assert env.cfg.info_bits == 1
    for sh in env._shards:
        sh.h.seed(ss[sh.lo:sh.hi])
    env.reset()
    flagged = np.zeros(n, np.int64)
    for t in range(T):
        obs, rew, dones, infos = env.step(actions[t])
        assert dones.dtype == bool
        for i, d in enumerate(infos):
            if d.get("ik_crossed"):
                flagged[i] += 1
    count = env.ik_crossed() >> 1
    assert flagged.sum() > 100 and np.array_equal(flagged, count), (flagged, count)
    env.close()
    env = HipVecEnv("KukaButtonGymEnv-v0", 256, seed=0, env_kwargs={"srl_model": "ground_truth"})
    env.reset()
    rs = np.random.RandomState(0)
    for t in range(300):
        _, _, _, infos = env.step(rs.randint(6, size=256))
        assert not any("ik_crossed" in d for d in infos)
    env.close()
    # the bit is opt-in at the ABI: a default handle's done bytes stay 0 / 1 under the same script
    cfg = _lib.default_config(_lib.ENV_KUKA_BUTTON)
    cfg.num_envs, cfg.rng_mode = n, _lib.RNG_MT19937
    h = _lib.Handle(cfg)
    h.seed(ss); h.reset()
    out = h.rollout(T, actions=actions[:T])
    assert set(np.unique(out["done"])) <= {0, 1} and (h.get_state(_lib.F_KUKA_IK_CROSSED) >> 1).sum() > 100
    h.close()


def test_dataset_generator_and_create_envs_take_device_ids(tmp_path):
    """environments/dataset_generator.py:166-190 forks num_cpu worker processes; here the lanes are spread over --device-ids (one handle
    per GPU): the dataset is the single-device dataset file for file.  rl_baselines/utils.py:194-229 createEnvs: args.device_ids."""
    import types
    from environments import dataset_generator as dg
    from rl_baselines.utils import createEnvs
    root = str(tmp_path) + "/"
    common = ["--num-cpu", "5", "--num-episode", "7", "--save-path", root, "--env", "MobileRobotGymEnv-v0", "--seed", "3", "--img-size", "32"]
    dg.main(common + ["--name", "one"])
    dg.main(common + ["--name", "many", "--device-ids", "0,0,0"])
    for fname in ("ground_truth.npz", "preprocessed_data.npz"):
        a, b = np.load(root + "one/" + fname), np.load(root + "many/" + fname)
        assert sorted(a.files) == sorted(b.files)
        for k in a.files:
            if k == "images_path":
                assert [p.split("/", 1)[1] for p in a[k]] == [p.split("/", 1)[1] for p in b[k]]
            else:
                assert np.array_equal(a[k], b[k]), k
    with open(root + "one/record_004/frame000017.jpg", "rb") as f, open(root + "many/record_004/frame000017.jpg", "rb") as g:
        assert f.read() == g.read()
    args = types.SimpleNamespace(env="KukaButtonGymEnv-v0", num_cpu=12, seed=4, srl_model="ground_truth", device_ids="0,0,0", num_stack=1)
    env = createEnvs(args)
    inner = env.unwrapped if hasattr(env, "unwrapped") else env
    while not hasattr(inner, "_shards"):
        inner = inner.venv
    assert len(inner._shards) == 3 and inner.num_envs == 12
    env.reset()
    obs, rew, done, infos = env.step(np.zeros(12, np.int64))
    assert obs.shape == (12, 3) and len(infos) == 12
    env.close()
    args.num_cpu, args.device_ids = 2, "0,0,0,0"                  # more devices than envs: one env per shard at most
    env = createEnvs(args)
    env.reset(); env.close()
