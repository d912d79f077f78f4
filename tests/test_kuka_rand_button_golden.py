"""KukaRandButtonGymEnv: the oracle's reset draws / distractor placement / shifted RNG stream against vectors produced by
the reference's own source (tests/golden/make_kuka_rand_button_golden.py).  Bit-exact."""
import os

import numpy as np
import pytest

from oracle import clib, kuka_clib


@pytest.fixture(scope="module")
def golden(golden_dir):
    clib.build()
    return np.load(os.path.join(golden_dir, "kuka_rand_button_reference.npz"))


def test_distractor_draws_and_following_commands_match_reference(golden):
    kuka_clib.set_variant(kuka_clib.VARIANT_RAND)
    try:
        n_cases = 0
        for seed in range(6):
            for rt in (0, 1):
                tag = "s{}|rt{}|".format(seed, rt)
                actions = golden[tag + "actions"].astype(np.int32)
                tr = kuka_clib.command_trace(seed, len(actions), actions, random_target=bool(rt), force_down=True)
                objs = kuka_clib.last_objects()
                kept = objs[objs[:, 2] > 0]
                assert len(kept) == int(golden[tag + "n_objects"]) and 0 < len(kept) <= 10
                ref = golden[tag + "object_pos"]
                assert np.array_equal(kept[:, :2], ref[:, :2]) and np.all(ref[:, 2] == -0.2 + 0.1)
                assert np.array_equal(kuka_clib.last_buttons()[:2], golden[tag + "button_xy"])
                assert np.array_equal(golden[tag + "ball_pos"], [0.25, -0.2, -0.2 + 0.3])
                # the env's RNG stream continues 20 draws later than KukaButtonGymEnv's: init actions and step noise
                assert np.array_equal(tr["reset_ee"][-5:], golden[tag + "reset_ik"]), tag
                n = tr["n_steps"]
                assert n == len(actions) and np.array_equal(tr["ee"][:n], golden[tag + "ik"][:n]), tag
                assert int(golden[tag + "n_pushes"]) == 1 and int(golden[tag + "max_steps"]) == 1000
                n_cases += 1
        assert n_cases == 12
    finally:
        kuka_clib.set_variant(kuka_clib.VARIANT_BUTTON)
