"""Pin the CPU oracle against vectors produced by the reference's own source
(tests/golden/make_mobile_golden.py -> mobile_reference.npz).  Bit-exact."""
import os

import numpy as np
import pytest

from oracle import gym_seeding, mobile_oracle


def test_gym_seeding_digits():
    # SURVEY.md App. B.1 / A.1 known answers
    assert gym_seeding.hash_seed_digits(0) == [547404849, 309914516]
    assert gym_seeding.hash_seed_digits(1) == [2739863373, 598274112]
    rng, _ = gym_seeding.np_random(0)
    assert rng.uniform(-4 / 3, 4 / 3) == -1.1883731833312448
    assert rng.uniform(-4 / 3, 4 / 3) == 1.2410424993128542
    rng, _ = gym_seeding.np_random(1)
    assert rng.uniform(-4 / 3, 4 / 3) == 0.8197078286103758


@pytest.fixture(scope="module")
def golden(golden_dir):
    return np.load(os.path.join(golden_dir, "mobile_reference.npz"))


def _cases(golden):
    tags = sorted({k.rsplit("|", 1)[0] for k in golden.files})
    assert len(tags) == 60
    return tags


def test_mobile_oracle_matches_reference_source(golden):
    for tag in _cases(golden):
        kind, s, rt, sr, c = tag.split("|")
        actions = golden[tag + "|actions"]
        out = mobile_oracle.rollout(mobile_oracle.KIND_NAMES[kind], int(s[1:]), actions,
                                    is_discrete=(c == "c0"), random_target=(rt == "rt1"),
                                    shape_reward=(sr == "sr1"))
        for field in ("obs0", "obs", "reward", "done", "pos", "target", "reset_obs"):
            ref = golden[tag + "|" + field]
            got = np.asarray(out[field]).reshape(ref.shape)
            assert np.array_equal(ref, got), (tag, field)


def test_mobile_invariants(golden):
    # episode length 251; bump margins; bump overrides reach reward (SURVEY §7 test plan)
    tag = "mobile|s0|rt0|sr0|c0"
    done = golden[tag + "|done"]
    assert list(np.nonzero(done)[0][:2]) == [250, 501]
    pos = golden[tag + "|pos"]
    assert pos[:, 0].min() >= 0.425 and pos[:, 0].max() <= 3.575
    assert pos[:, 1].min() >= 0.2 and pos[:, 1].max() <= 3.8
    assert set(np.unique(golden[tag + "|reward"])) <= {-1.0, 0.0, 1.0}


def test_norm2_matches_numpy():
    rs = np.random.RandomState(0)
    for v in rs.uniform(-3, 3, (20000, 2)):
        assert mobile_oracle.norm2(v[0], v[1]) == np.linalg.norm(v, 2)
    for v in rs.uniform(-1, 1, (20000, 3)):
        assert mobile_oracle.norm2(v[0], v[1], v[2]) == np.linalg.norm(v, 2)
