"""The plain-C oracle (oracle/*.c) against numpy's RandomState, the Python
restatement and the reference-source golden vectors.  Bit-exact."""
import os

import numpy as np
import pytest

from oracle import clib, gym_seeding, mobile_oracle


@pytest.fixture(scope="module", autouse=True)
def _build():
    clib.build()


def test_c_mt19937_matches_numpy_randomstate():
    for seed in (0, 1, 12345, 2 ** 40 + 7):
        key = gym_seeding.hash_seed_digits(seed)
        u, g, r3 = clib.np_random_draws(key, 3000)
        rs = np.random.RandomState()
        rs.seed(key)
        assert np.array_equal(u, rs.random_sample(3000))
        assert np.array_equal(g, np.array([rs.normal(0.0, 1.0) for _ in range(3000)]))
        assert np.array_equal(r3, np.array([rs.randint(3) for _ in range(3000)], dtype=np.uint32))


def test_c_mobile_oracle_matches_reference_golden(golden_dir):
    golden = np.load(os.path.join(golden_dir, "mobile_reference.npz"))
    tags = sorted({k.rsplit("|", 1)[0] for k in golden.files})
    for tag in tags:
        kind, s, rt, sr, c = tag.split("|")
        actions = golden[tag + "|actions"]
        T = len(actions)
        disc = c == "c0"
        out = clib.mobile_rollout(mobile_oracle.KIND_NAMES[kind], [int(s[1:])], T,
                                  actions=actions.reshape((T, 1) + actions.shape[1:]), is_discrete=disc,
                                  random_target=(rt == "rt1"), shape_reward=(sr == "sr1"))
        ref_obs = golden[tag + "|obs"].astype(np.float32)      # VecEnv buffer cast
        done = golden[tag + "|done"]
        # on done steps the VecEnv returns the first obs of the next episode
        ref_obs[np.nonzero(done)[0]] = golden[tag + "|reset_obs"].astype(np.float32)
        assert np.array_equal(out["obs0"][0], golden[tag + "|obs0"].astype(np.float32)), tag
        assert np.array_equal(out["obs"][:, 0, :], ref_obs.reshape(T, -1)), tag
        assert np.array_equal(out["reward64"][:, 0], golden[tag + "|reward"]), tag
        assert np.array_equal(out["done"][:, 0].astype(bool), done), tag


def test_c_mobile_oracle_matches_python_oracle_random_seeds():
    rs = np.random.RandomState(7)
    for kind in range(4):
        n_act = 2 if kind == 1 else 4
        seeds = rs.randint(0, 10 ** 6, size=5)
        T = 300
        actions = rs.randint(n_act, size=(T, len(seeds))).astype(np.int32)
        out = clib.mobile_rollout(kind, seeds, T, actions=actions, random_target=True)
        for j, seed in enumerate(seeds):
            py = mobile_oracle.rollout(kind, int(seed), actions[:, j], random_target=True)
            assert np.array_equal(out["reward64"][:, j], py["reward"])
            assert np.array_equal(out["done"][:, j].astype(bool), py["done"])
