"""The PyBullet pin of the Kuka dynamics (north star: joint positions within 1e-4 of the reference PyBullet step, discrete
reward / done flags bit-exact).  The fixture tests/golden/kuka_pybullet_reference.npz is produced by
tests/golden/make_kuka_pybullet_golden.py on a machine with pybullet==1.8.6 (this repo's build container has none): until
it exists these tests SKIP — "PARITY UNPINNED" — and the dynamics parity claim stays GPU == oracle only.

The fixture's table also carries the solver details the pin decides as DATA (solver_detail bits, contact_erp, limit_erp,
linear_slop: tests/golden/fit_kuka_pin.py searches them on the oracle and writes the winner back with --write); fixtures written
before that section existed are padded with the defaults.

When the fixture exists: its FULL model table (srlhip_kuka_tree_model: the 12-DoF arm + gripper tree — link frames, inertial
parameters, limits, friction read from the loaded PyBullet body, scene heights) is installed in the oracle and in the HIP
stepper, the recorded seeds / actions are replayed, and every recorded step is compared (arm AND gripper joints)."""
import os

import numpy as np
import pytest

from oracle import kuka_clib

FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kuka_pybullet_reference.npz")
TOL = 1e-4


def load_fixture():
    if not os.path.exists(FIXTURE):
        pytest.skip("PARITY UNPINNED: tests/golden/kuka_pybullet_reference.npz is absent — run "
                    "tests/golden/make_kuka_pybullet_golden.py where PyBullet is installed and commit its output")
    return np.load(FIXTURE)


def fixture_table(fx):
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import fit_kuka_pin
    return fit_kuka_pin.table_of(fx)


def episodes_of(fx):
    """(seed, actions [T], slice into the recorded arrays) per recorded env: consecutive episodes of one seed are one rollout"""
    for seed in np.unique(fx["seed"]):
        idx = np.nonzero(fx["seed"] == seed)[0]
        yield int(seed), fx["action"][idx].astype(np.int32), idx


GRIPPER_JOINTS = [7, 8, 10, 11, 13]             # the movable gripper joints of kuka_with_gripper2.sdf (9 and 12 are fixed)


def compare(fx, idx, q, reward, done, gq=None, crossed=None):
    """crossed: the per-step IK conditioning flag of the replay (oracle ik_crossed / SRLHIP_F_KUKA_IK_CROSSED & 1).  From the first
    flagged step on, this env's record is not compared: no two float64 implementations agree behind such a crossing
    (tests/kuka_scripts.py; the random-action fixture is not expected to contain one)."""
    n_use = len(idx) if crossed is None or not np.any(crossed) else int(np.argmax(np.asarray(crossed) != 0))
    idx, q, reward, done = idx[:n_use], q[:n_use], reward[:n_use], done[:n_use]
    gq = None if gq is None else gq[:n_use]
    assert np.array_equal(done.astype(int), fx["done"][idx]), "done flags differ from PyBullet"
    assert np.array_equal(reward, fx["reward"][idx].astype(reward.dtype)), "rewards differ from PyBullet"
    live = fx["done"][idx] == 0                     # the state after a terminal step already belongs to the next episode
    err = np.abs(q[live] - fx["q"][idx][live]).max()
    assert err <= TOL, "max |q - q_pybullet| = {:.3e}".format(err)
    if gq is not None:
        gerr = np.abs(gq[live] - fx["q14"][idx][live][:, GRIPPER_JOINTS]).max()
        assert gerr <= TOL, "max |q_gripper - q_pybullet| = {:.3e}".format(gerr)
    return err


def test_oracle_matches_pybullet():
    fx = load_fixture()
    try:
        kuka_clib.set_tree_model(fixture_table(fx))               # switches the oracle to the full model with PyBullet's own numbers
        for seed, actions, idx in episodes_of(fx):
            out = kuka_clib.rollout([seed], len(actions), actions=actions[:, None], aux=True, ik_trace=True)
            compare(fx, idx, out["q"][:, 0], out["reward64"][:, 0], out["done"][:, 0], gq=out["q_all"][:, 0, 7:12], crossed=out["ik_crossed"][:, 0])
    finally:
        kuka_clib.set_full(False)


@pytest.mark.gpu
def test_hip_stepper_matches_pybullet():
    from srlhip import _lib
    fx = load_fixture()
    for seed, actions, idx in episodes_of(fx):
        cfg = _lib.default_config(_lib.ENV_KUKA_BUTTON)
        cfg.num_envs, cfg.seed0, cfg.rng_mode, cfg.auto_reset = 1, seed, _lib.RNG_MT19937, 1
        assert cfg.kuka_model == _lib.KUKA_MODEL_FULL
        h = _lib.Handle(cfg)
        h.set_kuka_tree_model(fixture_table(fx))                   # model AND solver details from the fixture
        h.reset()
        q, gq, rew, done, crossed = [], [], [], [], []
        for a in actions:
            crossed.append(int(h.get_state(_lib.F_KUKA_IK_CROSSED)[0]) & 1)   # read BEFORE the step: an auto-reset clears the bit
            o, r, d = h.step(np.array([a], np.int32))
            q.append(h.get_state(_lib.F_KUKA_Q)[:, 0].copy()); gq.append(h.get_state(_lib.F_KUKA_GRIPPER_Q)[:, 0].copy())
            rew.append(float(r[0])); done.append(int(d[0]))
        compare(fx, idx, np.array(q), np.array(rew), np.array(done), gq=np.array(gq), crossed=np.array(crossed[1:] + [0]))
        h.close()
