"""Pin the oracle's restatement of Kuka2ButtonGymEnv's wrapper (reset draws, the two button positions, IK targets,
goal switching, reward / termination bookkeeping) against vectors produced by the reference's own source
(kuka_2button_gym_env.py + kuka.py) driven by a scripted fake pybullet —
tests/golden/make_kuka_2button_golden.py.  Bit-exact."""
import os

import numpy as np
import pytest

from oracle import clib, kuka_clib


@pytest.fixture(scope="module")
def golden(golden_dir):
    clib.build()
    return np.load(os.path.join(golden_dir, "kuka_2button_reference.npz"))


@pytest.fixture(autouse=True)
def two_button_variant():
    kuka_clib.set_variant(kuka_clib.VARIANT_TWO)
    yield
    kuka_clib.set_variant(kuka_clib.VARIANT_BUTTON)


def tags(golden, prefix):
    return sorted({k.rsplit("|", 1)[0] for k in golden.files if k.startswith(prefix)})


def test_reset_draws_button_positions_and_commands_match_reference(golden):
    cases = tags(golden, "act|")
    assert len(cases) == 12
    for tag in cases:
        _, mode, s, rt = tag.split("|")
        seed, random_target = int(s[1:]), rt == "rt1"
        # constructor defaults of the variant (kuka_2button_gym_env.py:29-33) and the IK call signature (kuka.py:147-148:
        # four positional null-space lists, no jointDamping keyword)
        assert float(golden[tag + "|max_distance"]) == 2.0 and not bool(golden[tag + "|force_down"])
        assert int(golden[tag + "|max_steps"]) == 1500 and int(golden[tag + "|ik_extra_positional_args"]) == 4
        assert int(golden[tag + "|n_reset_sim"]) == 505
        actions = golden[tag + "|actions"]
        n_ref = int(golden[tag + "|n_steps"])
        assert n_ref == 1501                                   # counter > 1500
        tr = kuka_clib.command_trace(seed, len(actions), actions.astype(np.float32 if mode != "discrete" else np.int32),
                                     is_discrete=(mode == "discrete"), random_target=random_target, force_down=False)
        ref_pos = golden[tag + "|button_all_pos"]
        assert np.array_equal(kuka_clib.last_buttons(), ref_pos[:, :2].reshape(4)), tag
        assert np.all(ref_pos[:, 2] == -0.2 + 0.28)
        n = min(tr["n_steps"], n_ref)
        assert n > 100, tag
        assert np.array_equal(tr["reset_ee"][-5:], golden[tag + "|reset_ik"]), tag
        assert np.array_equal(tr["ee"][:n], golden[tag + "|ik"][:n]), tag


def test_goal_switching_reward_and_termination_match_reference(golden):
    cases = tags(golden, "rew|")
    assert len(cases) == 16
    saw_switch = saw_second_done = saw_table = saw_25 = saw_50 = False
    for tag in cases:
        _, s, sr, m = tag.split("|")
        shape_reward, max_distance = sr == "sr1", float(m[1:])
        grip, ct = golden[tag + "|gripper"], golden[tag + "|contact_table"]
        cb = (golden[tag + "|contact_b1"], golden[tag + "|contact_b2"])
        all_pos, sim_idx = golden[tag + "|all_pos"], golden[tag + "|sim_idx"]
        state = np.zeros(6)
        for t in range(len(sim_idx)):
            k = int(sim_idx[t])
            goal = int(state[4])
            state, reward, done = kuka_clib.wrapper_step_two(state, grip[k], all_pos, cb[goal][k], ct[k], shape_reward,
                                                             max_distance)
            assert reward == golden[tag + "|reward"][t], (tag, t)
            assert done == golden[tag + "|done"][t], (tag, t)
            assert list(state) == [golden[tag + "|counter"][t], golden[tag + "|n_contacts0"][t],
                                   golden[tag + "|n_outside"][t], golden[tag + "|terminated"][t],
                                   golden[tag + "|goal_id"][t], golden[tag + "|n_contacts1"][t]], (tag, t)
            assert np.array_equal(golden[tag + "|button_pos"][t], all_pos[int(state[4])])
            # ground-truth observation = gripper - button_pos of the (possibly just switched) goal
            assert np.array_equal(golden[tag + "|obs"][t], grip[k] - all_pos[int(state[4])])
            saw_25 |= reward == 25
            saw_50 |= reward == 50
        saw_switch |= golden[tag + "|goal_id"][-1] == 1
        saw_second_done |= golden[tag + "|n_contacts1"][-1] >= 5
        saw_table |= bool(ct[int(sim_idx[-1])])
    assert saw_switch and saw_second_done and saw_25 and saw_50
