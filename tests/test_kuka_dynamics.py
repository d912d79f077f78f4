"""Cross-check the C oracle's Featherstone ABA / FK / IK against an independent
numpy Lagrangian formulation (tests/kuka_numpy_ref.py), plus model invariants."""
import numpy as np
import pytest

import kuka_numpy_ref as ref
from oracle import clib, kuka_clib


@pytest.fixture(scope="module", autouse=True)
def _build():
    clib.build()


def test_forward_kinematics_matches_numpy():
    rs = np.random.RandomState(0)
    for _ in range(20):
        q = rs.uniform(-2, 2, 7)
        R, p = kuka_clib.fk(q)
        R2, p2 = ref.fk(q)
        assert np.allclose(R, R2, atol=1e-14) and np.allclose(p, p2, atol=1e-14)


def test_aba_matches_lagrangian_forward_dynamics():
    rs = np.random.RandomState(1)
    for _ in range(25):
        q, qd, tau = rs.uniform(-2, 2, 7), rs.uniform(-1.5, 1.5, 7), rs.uniform(-20, 20, 7)
        qdd = kuka_clib.aba(q, qd, tau)
        expect = ref.forward_dynamics(q, qd, tau)
        assert np.allclose(qdd, expect, rtol=2e-6, atol=2e-6), np.abs(qdd - expect).max()


def test_minv_is_inverse_of_jacobian_mass_matrix():
    rs = np.random.RandomState(2)
    for _ in range(10):
        q = rs.uniform(-2, 2, 7)
        W = kuka_clib.minv(q)
        M = ref.mass_matrix(q)
        assert np.allclose(W, W.T, atol=1e-10)
        assert np.allclose(W @ M, np.eye(7), atol=1e-9)


def test_ik_step_reduces_error_and_settle_reaches_target():
    s = kuka_clib.settled()
    R, p = kuka_clib.fk(s["q"])
    ee = p[6] + R[6] @ [0, 0, 0.02]
    assert np.allclose(ee, [0.537, 0.0, 0.5], atol=1e-6)          # kuka.py:73
    assert np.allclose(R[6][:, 2], [0, 0, -1], atol=1e-5)          # flange pointing down (kuka.py:144)
    assert np.abs(s["qd"]).max() < 1e-3
    q_des = kuka_clib.ik(s["q"], [0.55, 0.05, 0.45])
    assert np.abs(q_des - s["q"]).max() < np.pi / 4 + 1e-12
    R2, p2 = kuka_clib.fk(q_des)
    ee2 = p2[6] + R2[6] @ [0, 0, 0.02]
    assert np.linalg.norm(ee2 - [0.55, 0.05, 0.45]) < 0.3 * np.linalg.norm(ee - [0.55, 0.05, 0.45])
