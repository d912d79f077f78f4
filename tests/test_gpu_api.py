"""GPU API-surface tests: per-env gym façade, HipVecEnv under the createEnvs wrappers,
random-agent rollouts and the dataset_generator CLI — the numeric twins of the
reference's exit-code smoke tests (tests/test_pipeline.py:95-111,
tests/test_dataset_manipulation.py:17-43)."""
import os
import types

import numpy as np
import pytest

from oracle import kuka_clib, mobile_oracle

pytestmark = pytest.mark.gpu


def test_single_env_facade_matches_python_oracle_exactly():
    """env.seed(s); env.reset(); env.step(a): float64 observations, int rewards — identical to the
    restated reference (which is pinned to the reference source's golden vectors)."""
    from environments.mobile_robot.mobile_robot_env import MobileRobotGymEnv
    from environments.mobile_robot.mobile_robot_2target_env import MobileRobot2TargetGymEnv
    for cls, kind in ((MobileRobotGymEnv, mobile_oracle.MOBILE), (MobileRobot2TargetGymEnv, mobile_oracle.MOBILE_2TARGET)):
        env = cls(srl_model="ground_truth", random_target=True, unknown_kwarg=1)
        ora = mobile_oracle.MobileOracleEnv(kind, random_target=True)
        assert env.action_space.n == 4 and env.observation_space.shape == (2,)
        assert env.render().size == 0
        for seed in (5, 6):
            assert env.seed(seed) == [seed]
            ora.seed(seed)
            assert np.array_equal(env.reset(), ora.reset())
            arng = np.random.RandomState(seed)
            for _ in range(260):
                a = arng.randint(4)
                obs, r, d, info = env.step(a)
                o2, r2, d2 = ora.step(a)
                assert np.array_equal(obs, o2) and r == r2 and type(r) is int and d == d2 and info == {}
                assert np.array_equal(env.getGroundTruth(), ora.ground_truth())
                assert np.array_equal(env.getTargetPos(), ora.target_pos())
            assert d
        env.close()


def test_kuka_facade_against_oracle():
    from environments.kuka_gym.kuka_button_gym_env import KukaButtonGymEnv
    env = KukaButtonGymEnv(srl_model="joints_position", max_distance=0.28)
    assert env.observation_space.shape == (17,) and env.action_space.n == 6
    env.seed(3)
    obs = env.reset()
    T = 420
    actions = np.random.RandomState(3).randint(6, size=T)
    ora = kuka_clib.rollout([3], T, actions=actions.reshape(T, 1).astype(np.int32), max_distance=0.28, obs_mode=2,
                            auto_reset=False)
    assert np.abs(obs - ora["obs0"][0]).max() < 1e-4 and obs.dtype == np.float64
    for t in range(T):
        obs, r, d, _ = env.step(int(actions[t]))
        assert np.abs(obs - ora["obs"][t, 0]).max() < 1e-4
        assert r == ora["reward64"][t, 0] and d == bool(ora["done"][t, 0])
        assert np.abs(np.array(env.getArmPos()) - ora["gripper"][t, 0]).max() < 1e-4
        if d:
            break
    assert len(env._kuka.joint_positions) == 14 and env.getTargetPos().shape == (3,)
    none_obs, r, d, _ = env.step(None)          # `None` action: zero motor command
    assert none_obs.shape == (17,)
    env.close()


@pytest.mark.parametrize("env_id", ["KukaButtonGymEnv-v0", "KukaMovingButtonGymEnv-v0", "Kuka2ButtonGymEnv-v0", "KukaRandButtonGymEnv-v0",
                                    "MobileRobotGymEnv-v0", "MobileRobot2TargetGymEnv-v0",
                                    "MobileRobot1DGymEnv-v0", "MobileRobotLineTargetGymEnv-v0"])
def test_random_agent_through_createEnvs(env_id, tmp_path):
    """1600 steps x 4 envs ("long enough to call a reset"), ground_truth, Monitor CSVs written."""
    from rl_baselines.random_agent import RandomAgentModel
    args = types.SimpleNamespace(env=env_id, num_cpu=4, seed=0, log_dir=str(tmp_path), num_stack=1, srl_model="ground_truth",
                                 num_timesteps=1600)
    seen = []

    def callback(_locals, _globals):
        seen.append((_locals["obs"].shape, _locals["reward"].dtype, _locals["done"].dtype))
    RandomAgentModel().train(args, callback, env_kwargs={"srl_model": "ground_truth", "is_discrete": True})
    assert len(seen) == 400 and seen[0][1] == np.float32 and seen[0][2] == np.bool_
    for i in range(4):
        lines = open(os.path.join(str(tmp_path), "%d.monitor.csv" % i)).read().splitlines()
        assert lines[0].startswith("#{") and lines[1] == "r,l,t"
        if "Mobile" in env_id:
            assert len(lines) == 3 and lines[2].split(",")[1] == "251"      # one 251-step episode in 400 steps


def test_vec_env_semantics_none_actions_and_infos():
    from srlhip.vec_env import HipVecEnv
    env = HipVecEnv("MobileRobotGymEnv-v0", 8, seed=3, env_kwargs={"srl_model": "ground_truth"})
    obs = env.reset()
    assert obs.shape == (8, 2) and obs.dtype == np.float32
    for t in range(251):
        obs, rew, done, infos = env.step([None if i % 2 else 1 for i in range(8)])
    assert done.all() and all(info["episode"]["l"] == 251 for info in infos)
    kept = infos
    _, _, _, infos2 = env.step([1] * 8)                # SubprocVecEnv hands out a fresh tuple per step: a kept list is not mutated
    assert infos2 is not kept and all("episode" in info for info in kept) and all(info == {} for info in infos2)
    ora = mobile_oracle.MobileOracleEnv(mobile_oracle.MOBILE)
    ora.seed(3 + 0)
    for _ in range(252):
        pass
    env.close()


def test_dataset_generator_cli(tmp_path):
    """--num-cpu 4 --num-episode 8 --force, with and without recording; seeds as the reference."""
    from environments import dataset_generator as dg
    root = str(tmp_path) + "/"
    common = ["--num-cpu", "4", "--num-episode", "8", "--save-path", root, "--env", "MobileRobotGymEnv-v0", "--seed", "1"]
    dg.main(common + ["--name", "mob", "--no-record-data"])
    assert not os.path.exists(root + "mob")
    dg.main(common + ["--name", "mob", "--force", "--reward-dist", "--img-size", "64"])
    assert sorted(os.listdir(root + "mob"))[:4] == ["dataset_config.json", "env_globals.json", "ground_truth.npz", "preprocessed_data.npz"]
    gt, pp = np.load(root + "mob/ground_truth.npz"), np.load(root + "mob/preprocessed_data.npz")
    assert len(pp["rewards"]) == 8 * 251 and pp["episode_starts"].sum() == 8 and gt["target_positions"].shape == (8, 2)
    from PIL import Image
    frame = np.asarray(Image.open(root + str(gt["images_path"][5]) + ".jpg"))
    assert frame.shape == (64, 64, 3) and frame.std() > 10            # a rendered arena, written as JPEG
    assert len(os.listdir(root + "mob/record_000")) == 251
    # episode k of the merged dataset == oracle episode seeded base+k with the action-space stream of the same seed
    base = np.random.RandomState(1).randint(int(1e10))
    for k in (0, 3, 7):
        env = mobile_oracle.MobileOracleEnv(mobile_oracle.MOBILE)
        env.seed(base + k)
        arng = np.random.RandomState((base + k) % 2 ** 32)
        env.reset()
        sl = slice(k * 251, (k + 1) * 251)
        assert np.array_equal(gt["ground_truth_states"][sl][0], env.ground_truth())
        for t in range(251):
            a = arng.randint(4)
            _, r, d = env.step(a)
            assert pp["actions"][sl][t] == a and pp["rewards"][sl][t] == r
            if t < 250:
                assert np.array_equal(gt["ground_truth_states"][sl][t + 1], env.ground_truth())
    with pytest.raises(AssertionError):
        dg.main(common + ["--name", "mob"])              # exists, no --force
    dg.main(["--num-cpu", "2", "--num-episode", "3", "--save-path", root, "--name", "kuka", "--env", "KukaButtonGymEnv-v0",
             "--img-size", "64"])
    pp = np.load(root + "kuka/preprocessed_data.npz")
    assert pp["episode_starts"].sum() == 3 and set(np.unique(pp["rewards"])) <= {-1, 0, 1}


def test_episode_records_are_the_episode_stats_without_a_copy(tmp_path):
    """srlhip_episode_records: host-pointer handles keep Monitor's (r, l) in mapped host memory the kernels write — the views equal
    srlhip_episode_stats' copies after per-step calls, fused rollouts and masked resets, on both env families; HipVecEnv's
    info['episode'] (read from the views) matches, with buffered Monitor rows flushed on close; device-pointer handles refuse."""
    from srlhip import _lib
    from srlhip.vec_env import HipVecEnv
    for kind, T in ((_lib.ENV_KUKA_BUTTON, 1300), (_lib.ENV_MOBILE, 600)):
        cfg = _lib.default_config(kind)
        cfg.num_envs, cfg.rng_mode, cfg.auto_reset = 96, _lib.RNG_PHILOX, 1
        h = _lib.Handle(cfg)
        h.reset()
        ret_v, len_v = h.episode_records()
        acts = np.random.RandomState(1).randint(4 if kind == _lib.ENV_MOBILE else 6, size=(T, 96)).astype(np.int32)
        seen = 0
        for t in range(T // 2):
            o, r, d = h.step(acts[t])
            seen += int(d.sum())
        h.rollout(T - T // 2, actions=acts[T // 2:])
        ret, length, fin = h.episode_stats()
        assert fin.sum() >= seen and fin.min() >= 1 and np.array_equal(ret, ret_v) and np.array_equal(length, len_v)
        assert np.array_equal(h.get_state(_lib.F_LAST_RETURN), ret_v) and np.array_equal(h.get_state(_lib.F_LAST_LENGTH), len_v)
        h.close()
    cfg = _lib.default_config(_lib.ENV_MOBILE)
    cfg.num_envs, cfg.io_device, cfg.rng_mode = 8, 1, _lib.RNG_PHILOX
    h = _lib.Handle(cfg)
    with pytest.raises(_lib.SrlHipError):
        h.episode_records()
    h.close()
    env = HipVecEnv("KukaButtonGymEnv-v0", 128, seed=0, env_kwargs={"srl_model": "ground_truth"}, log_dir=str(tmp_path))
    env.reset()
    rs, n_ep, ret_sum = np.random.RandomState(2), np.zeros(128, int), np.zeros(128)
    running = np.zeros(128)
    for t in range(1200):
        obs, rew, done, infos = env.step(rs.randint(6, size=128))
        running += rew
        for i in np.flatnonzero(done):
            assert infos[i]["episode"]["r"] == round(float(running[i]), 6)
            n_ep[i] += 1; running[i] = 0.0
        assert all(info == {} for i, info in enumerate(infos) if not done[i])
    env.close()
    for i in (0, 77, 127):
        rows = open(os.path.join(str(tmp_path), "%d.monitor.csv" % i)).read().splitlines()[2:]
        assert len(rows) == n_ep[i] >= 1
