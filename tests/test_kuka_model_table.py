"""The recalled part of the Kuka model is runtime DATA (srlhip_kuka_model, 138 doubles): the same table drives the oracle,
the lane-group stepper (CPU emulation here, the device in test_gpu_kuka_model.py) and can be read from an sdf file."""
import os

import numpy as np
import pytest

import hostcheck
from oracle import clib, kuka_clib
from srlhip import kuka_model


@pytest.fixture(scope="module", autouse=True)
def _build():
    clib.build()
    hostcheck.lib()


def perturbed(m0, k=1):
    m = {name: np.array(v, copy=True) if not np.isscalar(v) else v for name, v in m0.items()}
    rs = np.random.RandomState(k)
    m["mass"] = m0["mass"] * rs.uniform(0.8, 1.3, 7)
    m["com"] = m0["com"] + rs.uniform(-0.01, 0.01, (7, 3))
    m["inertia"] = m0["inertia"] * rs.uniform(0.8, 1.3, (7, 3))
    m["joint_xyz"] = m0["joint_xyz"] * 1.05
    m["joint_rpy"] = m0["joint_rpy"] + rs.uniform(-0.03, 0.03, (7, 3))
    m["joint_damping"] = 0.7
    m["gripper_point"] = m0["gripper_point"] + np.array([0.0, 0.01, 0.02])
    m["sphere"] = m0["sphere"] * 1.1
    return m


def test_baked_tables_agree():
    """library (csrc/kuka_core.hpp), oracle (oracle/kuka_model.h) and the host harness hold the same baked model"""
    lib, ora, host = kuka_model.default(), kuka_clib.get_model(), hostcheck.default_model()
    for name, _ in kuka_model.MODEL_FIELDS:
        assert np.array_equal(np.asarray(lib[name]), np.asarray(ora[name])), name
        assert np.array_equal(np.asarray(lib[name]), np.asarray(host[name])), name
    assert np.array_equal(kuka_model.to_table(kuka_model.to_dict(kuka_model.to_table(lib))), kuka_model.to_table(lib))


def test_runtime_model_drives_oracle_and_lane_group_stepper():
    n, T = 3, 450
    actions = np.random.RandomState(0).randint(6, size=(T, n)).astype(np.int32)
    actions[:, 0] = 4                                            # presses down: contacts with the (rescaled) gripper spheres
    m0 = kuka_clib.get_model()
    base = kuka_clib.rollout(np.arange(n), T, actions=actions)
    try:
        # the runtime-table code path with the baked table reproduces the baked model
        hostcheck.set_model(m0)
        b = hostcheck.group_rollout(np.arange(n), T, actions=actions)
        assert np.abs(base["q"] - b["q"]).max() < 1e-8 and np.array_equal(base["done"], b["done"]) and np.array_equal(base["reward"], b["reward"])
        # a different arm: both implementations follow it, and it is a different arm
        m = perturbed(m0)
        kuka_clib.set_model(m); hostcheck.set_model(m)
        a2 = kuka_clib.rollout(np.arange(n), T, actions=actions)
        b2 = hostcheck.group_rollout(np.arange(n), T, actions=actions)
        assert np.abs(a2["q"] - b2["q"]).max() < 1e-8
        assert np.array_equal(a2["done"], b2["done"]) and np.array_equal(a2["reward"], b2["reward"])
        assert np.abs(a2["obs"] - b2["obs"]).max() < 1e-6
        assert np.abs(a2["q"] - base["q"]).max() > 1e-2
    finally:
        kuka_clib.set_model(m0); hostcheck.set_model(None)


def write_sdf(path, m, gripper_mass=0.0):
    """a minimal sdf with the structure of kuka_with_gripper2.sdf: absolute link poses, inertial pose / mass / diagonal inertia"""
    T = np.eye(4)
    links, joints = [], []
    links.append('<link name="lbr_iiwa_link_0"><pose>0 0 0 0 0 0</pose><inertial><pose>0 0 0 0 0 0</pose><mass>0</mass>'
                 '<inertia><ixx>1</ixx><iyy>1</iyy><izz>1</izz></inertia></inertial></link>')
    for i in range(7):
        J = np.eye(4)
        J[:3, :3], J[:3, 3] = kuka_model._rpy_matrix(m["joint_rpy"][i]), m["joint_xyz"][i]
        T = T @ J
        rpy = kuka_model._matrix_rpy(T[:3, :3])
        mass, com, I = m["mass"][i], m["com"][i], m["inertia"][i]
        if i == 6 and gripper_mass:
            mass = mass - gripper_mass
        links.append('<link name="lbr_iiwa_link_{}"><pose>{} {} {} {} {} {}</pose><inertial><pose>{} {} {} 0 0 0</pose><mass>{}</mass>'
                     '<inertia><ixx>{}</ixx><iyy>{}</iyy><izz>{}</izz></inertia></inertial></link>'.format(
                         i + 1, *T[:3, 3], *rpy, *com, repr(float(mass)), *I))
        joints.append('<joint name="J{}" type="revolute"><parent>lbr_iiwa_link_{}</parent><child>lbr_iiwa_link_{}</child><axis><xyz>0 0 1</xyz>'
                      '<limit><lower>{}</lower><upper>{}</upper></limit><dynamics><damping>{}</damping></dynamics></axis></joint>'.format(
                          i, i, i + 1, repr(float(m["joint_lower"][i])), repr(float(m["joint_upper"][i])), repr(float(m["joint_damping"]))))
    if gripper_mass:
        com = m["com"][6]
        rpy = kuka_model._matrix_rpy(T[:3, :3])
        links.append('<link name="base_link"><pose>{} {} {} {} {} {}</pose><inertial><pose>{} {} {} 0 0 0</pose><mass>{}</mass>'
                     '<inertia><ixx>1e-9</ixx><iyy>1e-9</iyy><izz>1e-9</izz></inertia></inertial></link>'.format(
                         *T[:3, 3], *rpy, *com, repr(float(gripper_mass))))
        joints.append('<joint name="gripper_to_arm" type="fixed"><parent>lbr_iiwa_link_7</parent><child>base_link</child></joint>')
    with open(path, "w") as f:
        f.write('<sdf version="1.6"><world name="w"><model name="lbr_iiwa">{}{}</model></world></sdf>'.format("".join(links), "".join(joints)))


def test_model_table_from_sdf(tmp_path):
    m = perturbed(kuka_model.default(), k=3)
    path = str(tmp_path / "arm.sdf")
    write_sdf(path, m)
    got = kuka_model.from_sdf(path)
    for i in range(7):
        assert np.allclose(kuka_model._rpy_matrix(got["joint_rpy"][i]), kuka_model._rpy_matrix(m["joint_rpy"][i]), atol=1e-12)
    for name in ("joint_xyz", "joint_lower", "joint_upper", "mass", "com", "inertia"):
        assert np.allclose(got[name], m[name], atol=1e-12), name
    assert got["joint_damping"] == m["joint_damping"]
    # a gripper body behind link 7 is lumped into it (same centre of mass here: mass adds up, inertia grows by ~0)
    write_sdf(path, m, gripper_mass=0.4)
    lumped = kuka_model.from_sdf(path)
    assert np.isclose(lumped["mass"][6], m["mass"][6]) and np.allclose(lumped["com"][6], m["com"][6], atol=1e-12)
