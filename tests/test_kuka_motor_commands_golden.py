"""CPU: every p.setJointMotorControl2 call of the reference's Kuka wrapper against the model tables (round-3 verdict, item 2b).

tests/golden/kuka_wrapper_reference.npz `motorlog|<mode>|{reset,step0,step1}` holds ALL arguments of every call the reference's own
source makes (kuka.py:69-71 reset motors, kuka.py:167-187 applyAction, kuka_button_gym_env.py:347 button motor), recorded by
tests/golden/make_kuka_wrapper_golden.py from /root/reference with a scripted pybullet.  The product's baked
srlhip_kuka_tree_model (host call, no GPU), the oracle's table and the button-motor constants of both are compared with those
rows instead of with literals.  NaN in the fixture = the reference does not pass the argument; pybullet then applies its own
default, which is RECALLED here (positionGain 0.1, velocityGain 1.0, no velocity clamp, force 100000 for URDF-created motors) —
those four numbers are the only thing this file cannot take from the reference."""
import os
import re

import numpy as np
import pytest

from oracle import kuka_clib
from srlhip import _lib, kuka_model

HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, "golden", "kuka_wrapper_reference.npz"))
BODY, JOINT, MODE, TPOS, TVEL, FORCE, MAXVEL, PGAIN, VGAIN = range(9)
BUTTON_UID, KUKA_UID, POSITION_CONTROL = 2, 3, 2          # load order of the scripted pybullet; p.POSITION_CONTROL of the stub
PYBULLET_DEFAULT_POSITION_GAIN, PYBULLET_DEFAULT_VELOCITY_GAIN = 0.1, 1.0       # [RECALLED] pybullet 1.8.6 setJointMotorControl2 defaults


def tables():
    prod = kuka_model.tree_to_dict(_lib.kuka_tree_default_model())
    kuka_clib.set_full(True)
    try:
        ora = kuka_model.tree_to_dict(kuka_clib.get_tree_model())
    finally:
        kuka_clib.set_full(False)
    return {"product": prod, "oracle": ora}


def constant(path, name):
    src = open(os.path.join(HERE, "..", path)).read()
    m = re.search(r"\b" + name + r"\s*=?\s*(-?[0-9.eE+-]+)", src)
    assert m, (path, name)
    return float(m.group(1))


@pytest.mark.parametrize("mode", ["discrete", "continuous", "joints"])
def test_per_step_motor_commands_match_the_model_tables(mode):
    for step in ("step0", "step1"):
        log = G["motorlog|{}|{}".format(mode, step)]
        kuka = log[log[:, BODY] == KUKA_UID]
        assert np.all(log[:, MODE] == POSITION_CONTROL)
        # one command per motorised joint per applyAction, the seven arm joints first (kuka.py:167-170), then 7, 8, 11, 10, 13
        assert [int(j) for j in kuka[:, JOINT]] == [0, 1, 2, 3, 4, 5, 6, 7, 8, 11, 10, 13]
        for name, m in tables().items():
            by_joint = {int(j["joint_index"]): j for j in m["joints"]}
            assert sorted(by_joint) == sorted(int(j) for j in kuka[:, JOINT]), name      # every DoF of the table is commanded, nothing else is
            for row in kuka:
                j = by_joint[int(row[JOINT])]
                assert j["max_force"] == row[FORCE], (name, row)
                assert j["kp"] == (PYBULLET_DEFAULT_POSITION_GAIN if np.isnan(row[PGAIN]) else row[PGAIN]), (name, row)
                if np.isnan(row[MAXVEL]):
                    assert j["max_vel"] >= 1e29, (name, row)                              # no clamp
                else:
                    assert j["max_vel"] == row[MAXVEL], (name, row)
                # the kernels and the oracle integrate velocityGain 1 and targetVelocity 0 on every motor
                assert np.isnan(row[VGAIN]) or row[VGAIN] == PYBULLET_DEFAULT_VELOCITY_GAIN
                assert np.isnan(row[TVEL]) or row[TVEL] == 0.0
        # gripper targets: end_effector_angle (da = 0 in every env: 0.0), fingers -/+ finger_angle = -/+ 0.0, tips 0 — the ONLY
        # command the steppers implement (tenv_step passes finger_angle = 0.0; the oracle's motor[4] is never set)
        grip = {int(r[JOINT]): r[TPOS] for r in kuka[7:]}
        assert grip == {7: 0.0, 8: 0.0, 11: 0.0, 10: 0.0, 13: 0.0}


@pytest.mark.parametrize("mode", ["discrete", "continuous", "joints"])
def test_button_motor_command(mode):
    """step2 (kuka_button_gym_env.py:347): ONE position command per env step, before the action-repeat loop, target 0.1, nothing else
    passed -> pybullet's default gain and force (recalled constants of both implementations)."""
    for step in ("step0", "step1"):
        log = G["motorlog|{}|{}".format(mode, step)]
        btn = log[log[:, BODY] == BUTTON_UID]
        assert btn.shape[0] == 1 and int(btn[0, JOINT]) == 1 and np.array_equal(np.nonzero(log[:, BODY] == BUTTON_UID)[0], [0])
        assert np.all(np.isnan(btn[0, TVEL:]))
        assert constant("robotics-rl-srl_amd/csrc/kuka_core.hpp", "kButtonTarget") == btn[0, TPOS]
        assert constant("oracle/kuka_model.h", "KM_BUTTON_TARGET") == btn[0, TPOS]
        assert constant("robotics-rl-srl_amd/csrc/kuka_core.hpp", "kButtonKp") == PYBULLET_DEFAULT_POSITION_GAIN
        assert constant("oracle/kuka_model.h", "KM_BUTTON_KP") == PYBULLET_DEFAULT_POSITION_GAIN
    # reset() never commands the button: until the first step() it keeps the velocity motor pybullet creates with the body
    assert not np.any(G["motorlog|{}|reset".format(mode)][:, BODY] == BUTTON_UID)


@pytest.mark.parametrize("mode", ["discrete", "continuous", "joints"])
def test_reset_motor_commands(mode):
    """Kuka.reset (kuka.py:64-71): all 14 joints get resetJointState + a position motor at joint_positions with force 200; every
    motorised joint is re-commanded by applyAction BEFORE the first stepSimulation (so the reset motors never act), joints 9 and 12
    are never commanded again (fixed joints in the recalled SDF: the command is a no-op there).  Then 500 settle applyActions + 5
    init actions = 505 x 12 commands."""
    log = G["motorlog|{}|reset".format(mode)]
    n_sim = int(G["motorlog|{}|n_reset_sim".format(mode)])
    assert n_sim == 505 and log.shape[0] == 14 + 505 * 12
    first = log[:14]
    assert [int(j) for j in first[:, JOINT]] == list(range(14)) and np.all(first[:, BODY] == KUKA_UID)
    assert np.all(first[:, FORCE] == 200.0) and np.all(np.isnan(first[:, [TVEL, MAXVEL, PGAIN, VGAIN]]))
    for name, m in tables().items():
        q0 = {int(j["joint_index"]): i for i, j in enumerate(m["joints"])}
        assert set(range(14)) - set(q0) == {9, 12}, name
    # the 14 targets are the joint_positions list the steppers reset to (kuka.py:65-66)
    ref_q0 = first[:, TPOS]
    src = open(os.path.join(HERE, "..", "oracle", "kuka_model.h")).read()
    m = re.search(r"KM_JOINT_POSITIONS\[14\]\s*=\s*\{([^}]*)\}", src)
    assert m and np.array_equal(np.array([float(x) for x in m.group(1).split(",")]), ref_q0)
    src = open(os.path.join(HERE, "..", "robotics-rl-srl_amd", "csrc", "kuka_core.hpp")).read()
    m = re.search(r"kJointPositions\[14\]\s*=\s*\{([^}]*)\}", src)
    assert m and np.array_equal(np.array([float(x) for x in m.group(1).split(",")]), ref_q0)
    rest = log[14:].reshape(505, 12, 9)
    assert np.all(rest[:, :, JOINT] == np.array([0, 1, 2, 3, 4, 5, 6, 7, 8, 11, 10, 13]))
    assert np.all(rest[:, 7:, TPOS] == 0.0)                       # gripper closed during the settle and init steps too
    if mode == "joints":                                          # settle: joint_positions[:7] re-commanded (kuka_button_gym_env.py:243-244)
        assert np.all(rest[:500, :7, TPOS] == ref_q0[:7])
