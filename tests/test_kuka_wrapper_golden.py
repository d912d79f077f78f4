"""Pin the oracle's restatement of the reference's Kuka control / reward WRAPPER against
vectors produced by the reference's own source (kuka.py, kuka_button_gym_env.py) driven by a
scripted fake pybullet — tests/golden/make_kuka_wrapper_golden.py.  Bit-exact."""
import os

import numpy as np
import pytest

from oracle import clib, kuka_clib


@pytest.fixture(scope="module")
def golden(golden_dir):
    clib.build()
    return np.load(os.path.join(golden_dir, "kuka_wrapper_reference.npz"))


def tags(golden, prefix):
    return sorted({k.rsplit("|", 1)[0] for k in golden.files if k.startswith(prefix)})


def test_commands_issued_by_reset_and_step_match_reference(golden):
    """RNG draw order, action tables, noise arithmetic, force_down, None actions, workspace clipping."""
    cases = tags(golden, "act|")
    assert len(cases) == 30
    for tag in cases:
        _, mode, s, rt, fd = tag.split("|")
        seed, random_target, force_down = int(s[1:]), rt == "rt1", fd == "fd1"
        actions = golden[tag + "|actions"]
        n_ref = int(golden[tag + "|n_steps"])
        assert n_ref == 1001                                   # reference episode ends at counter > 1000
        assert int(golden[tag + "|n_reset_sim"]) == 505        # 500 settle + 5 init actions
        tr = kuka_clib.command_trace(seed, len(actions), actions.astype(np.float32 if mode != "discrete" else np.int32),
                                     is_discrete=(mode == "discrete"), action_joints=(mode == "joints"),
                                     random_target=random_target, force_down=force_down)
        n = min(tr["n_steps"], n_ref)                          # the oracle's real physics may end the episode earlier
        assert n > 150, tag
        if mode == "joints":
            assert np.array_equal(tr["reset_jt"][-5:], golden[tag + "|reset_motor"]), tag
            assert np.array_equal(tr["jt"][:n], golden[tag + "|motor"][:n]), tag
        else:
            assert np.array_equal(tr["reset_ee"][-5:], golden[tag + "|reset_ik"]), tag
            assert np.array_equal(tr["ee"][:n], golden[tag + "|ik"][:n]), tag


def test_reward_and_termination_bookkeeping_matches_reference(golden):
    cases = tags(golden, "rew|")
    assert len(cases) == 16
    saw_contact_end = saw_table_end = False
    for tag in cases:
        _, s, sr, d, m = tag.split("|")
        shape_reward, is_discrete, max_distance = sr == "sr1", d == "d1", float(m[1:])
        grip, cb, ct = golden[tag + "|gripper"], golden[tag + "|contact_button"], golden[tag + "|contact_table"]
        button_pos, sim_idx = golden[tag + "|button_pos"], golden[tag + "|sim_idx"]
        state = np.zeros(4)
        for t in range(len(sim_idx)):
            k = int(sim_idx[t])
            state, reward, done = kuka_clib.wrapper_step(state, grip[k], button_pos, cb[k], ct[k], shape_reward,
                                                         is_discrete, max_distance)
            assert reward == golden[tag + "|reward"][t], (tag, t)
            assert done == golden[tag + "|done"][t], (tag, t)
            assert list(state) == [golden[tag + "|counter"][t], golden[tag + "|n_contacts"][t],
                                   golden[tag + "|n_outside"][t], golden[tag + "|terminated"][t]], (tag, t)
        saw_contact_end |= golden[tag + "|n_contacts"][-1] >= 5
        saw_table_end |= bool(ct[int(sim_idx[-1])])
    assert saw_contact_end or saw_table_end


def test_moving_button_wrapper_matches_reference(golden):
    """KukaMovingButtonGymEnv (kuka_moving_button_gym_env.py): np_random.choice([-1, 1]) is drawn first, the target
    moves +-1 mm per step and bounces at |y| = 0.3, shaped reward uses the 50 / -250 / -d branch, limit 1500 steps."""
    cases = tags(golden, "mov|")
    assert len(cases) == 8
    kuka_clib.set_moving(True)
    try:
        for tag in cases:
            _, s, sr, rt = tag.split("|")
            seed, shape_reward, random_target = int(s[1:]), sr == "sr1", rt == "rt1"
            actions = golden[tag + "|actions"].astype(np.int32)
            assert int(golden[tag + "|n_steps"]) == 1501
            tr = kuka_clib.command_trace(seed, 200, actions[:200], random_target=random_target)
            assert np.array_equal(tr["reset_ee"][-5:], golden[tag + "|reset_ik"]), tag     # pins the draw order
            assert np.array_equal(tr["ee"][:200], golden[tag + "|ik"][:200]), tag
            if not random_target:       # button starts at y = 0: the oracle's target must follow the same track
                T = 40
                out = kuka_clib.rollout([seed], T, actions=actions[:T].reshape(T, 1), auto_reset=False, trace=False)
                assert out["final_state"][0, 23] == golden[tag + "|button_y"][T - 1], tag
            # reward / termination on the scripted physics, with the moving target
            grip, cb, sim_idx = golden[tag + "|gripper"], golden[tag + "|contact_button"], golden[tag + "|sim_idx"]
            bp0, by = golden[tag + "|button_pos0"], golden[tag + "|button_y"]
            state = np.zeros(4)
            for t in range(len(sim_idx)):
                k = int(sim_idx[t])
                state, reward, done = kuka_clib.wrapper_step(state, grip[k], [bp0[0], by[t], bp0[2]], cb[k], False,
                                                             shape_reward, True, 0.8)
                assert reward == golden[tag + "|reward"][t] and done == golden[tag + "|done"][t], (tag, t)
            assert state[0] == 1501
    finally:
        kuka_clib.set_moving(False)
