"""CPU: the FULL-model Kuka stepper's own source (csrc/kuka_tree.hpp under the fiber harness) against the oracle under SATURATING
scripted policies in the default config (small workspace box, MT19937, one env per script x seed) — the regime a random agent never
reaches (round-4 verdict, weak #1; tests/kuka_scripts.py).  Bar: 1e-7 rad on the arm joints and reward / done bit for bit on every
env-step BEFORE the IK conditioning flag (kuka.py:41-42,118-156 -> SRLHIP_F_KUKA_IK_CROSSED); both implementations raise the flag at
the same step; after it nothing is asserted, the measured divergence is printed.  A subset of the catalogue runs here (the harness
integrates ~500 env-steps per second); tests/test_gpu_kuka_ik_crossing.py runs all of it on the GPU."""
import numpy as np
import pytest

import hostcheck
import kuka_scripts
from oracle import kuka_clib


@pytest.fixture(autouse=True)
def full_oracle():
    kuka_clib.set_full(True)
    yield
    kuka_clib.set_full(False)


def run(scripts, seeds, T, **kw):
    names, ss, actions = kuka_scripts.batch(scripts, seeds)
    ora = kuka_clib.rollout(ss, T, actions=actions, rng_mode=kuka_clib.RNG_MT19937, auto_reset=False, ik_trace=True, **kw)
    got = hostcheck.tree_rollout(ss, T, actions=actions, rng_mode=kuka_clib.RNG_MT19937, auto_reset=False, ik_trace=True, **kw)
    st = kuka_scripts.compare(names, ora, got["q"], got["reward"], got["done"], got["ik_crossed"])
    assert np.array_equal(ora["ik_final"], got["ik_final"])               # sticky bit and flagged-step count, as the product stores them
    print(st)
    return st, ora


def test_the_verdicts_corner_script_crosses_the_elbow_singularity_and_is_flagged_before_it_diverges():
    """+x 300, -y 300, down: joint 3 walks through 0 at step ~565 with joint 5 on its limit; |dq| reaches 1e-3 and done moves by a few
    steps AFTER the flag; up to the flag (step ~510) the two implementations agree to 1e-11."""
    T = 1000
    scripts = {k: v for k, v in kuka_scripts.discrete_scripts(T).items() if k == "1_2_down"}
    st, ora = run(scripts, (7, 8, 9, 10), T)
    assert st["crossed"] == 4 and st["pre_max_dq"] < 1e-9
    first = kuka_scripts.first_index(ora["ik_crossed"] != 0)
    assert (first > 480).all() and (first < 540).all()
    # the regime is real: at least one of the four diverges by > 1e-4 after the flag (if this ever stops being true the flag is too eager)
    assert st["post_max_dq"] > 1e-4


def test_held_actions_and_a_non_crossing_corner():
    T = 700
    all_scripts = kuka_scripts.discrete_scripts(T)
    st, ora = run({k: all_scripts[k] for k in ("hold1", "hold3", "hold4", "0_3_down")}, (7, 9), T)
    crossed = ora["ik_final"][:, 0].reshape(4, 2)
    assert crossed[0].all() and not crossed[2].any() and not crossed[3].any()       # +x crosses; down / (-x, +y, down) never do


def test_saturated_continuous_and_joint_space_actions_never_cross():
    T = 600
    st, ora = run({k: v for k, v in kuka_scripts.continuous_scripts(T).items() if k in ("cont+1+1-1", "cont-1+1+1")}, (7,), T, is_discrete=False)
    assert st["crossed"] == 0
    st, ora = run({k: v for k, v in kuka_scripts.joint_scripts(T).items() if k in ("joints+", "joints+-")}, (7,), T, is_discrete=False, action_joints=True)
    assert st["crossed"] == 0 and (ora["ik_det"] > 1e299).all()           # no IK on the joint-space path
