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


# ---- the FULL model (12-DoF arm + gripper tree): the oracle's tree ABA against a numpy formulation built from the table alone
@pytest.fixture
def full_model():
    import kuka_numpy_tree_ref as tref
    kuka_clib.set_full(True)
    yield tref, tref.unpack(kuka_clib.get_tree_model())
    kuka_clib.set_full(False)


def test_full_model_kinematics_and_mass_matrix_match_numpy(full_model):
    tref, J = full_model
    rs = np.random.RandomState(3)
    for _ in range(8):
        q = np.concatenate([rs.uniform(-2, 2, 7), rs.uniform(-0.4, 0.4, 5)])
        R, p = kuka_clib.fk(q)
        R2, p2 = tref.fk(J, q)
        assert np.allclose(R, R2, atol=1e-13) and np.allclose(p, p2, atol=1e-13)
        W, M = kuka_clib.minv(q), tref.mass_matrix(J, q)
        assert W.shape == (12, 12) and np.allclose(W, W.T, atol=1e-9)
        assert np.allclose(W @ M, np.eye(12), atol=1e-8)
        # branch-induced sparsity: neither finger is an ancestor of the other, so M has no left-right block — M^-1 is dense
        assert np.abs(M[8:10, 10:12]).max() == 0 and np.abs(W[8:10, 10:12]).min() > 0
        assert np.allclose(M, M.T, atol=1e-12) and np.linalg.eigvalsh(M).min() > 0


def test_full_model_aba_matches_lagrangian_forward_dynamics(full_model):
    tref, J = full_model
    rs = np.random.RandomState(4)
    for _ in range(10):
        q = np.concatenate([rs.uniform(-2, 2, 7), rs.uniform(-0.4, 0.4, 5)])
        qd = np.concatenate([rs.uniform(-1.5, 1.5, 7), rs.uniform(-3, 3, 5)])
        tau = np.concatenate([rs.uniform(-20, 20, 7), rs.uniform(-2, 2, 5)])
        qdd = kuka_clib.aba(q, qd, tau)
        expect = tref.forward_dynamics(J, q, qd, tau)
        assert np.allclose(qdd, expect, rtol=5e-6, atol=5e-6), np.abs(qdd - expect).max()


def test_full_model_settled_gripper_is_closed_and_arm_on_target(full_model):
    # the 500 settle steps: the fingers start at -+0.3 rad (kuka.py:65-66) and their 2 / 2.5 N m motors close them to 0 (finger_angle = 0)
    out = kuka_clib.rollout([0], 1, actions=np.array([[-1]], np.int32), aux=True)
    qa = out["q_all"][0, 0]
    assert np.abs(qa[7]) < 1e-4 and np.abs(qa[[8, 10]]).max() < 0.02 and np.abs(qa[[9, 11]]).max() < 0.02


def test_solver_is_not_converged_after_150_sweeps_so_row_order_matters():
    """profiles/probes/kuka_bullet_detail_sensitivity.py in small: the oracle's sensitivity bits (NOT part of the parity definition)
    change the joints by far more than the 1e-4 bar within a few hundred contact-free steps — the reason DESIGN 4.2 lists Bullet's
    recalled sweep order as the first thing the PyBullet pin has to settle."""
    import ctypes
    lib = kuka_clib._lib()
    lib.kuka_oracle_set_detail.argtypes = [ctypes.c_int]
    T = 150
    actions = np.random.RandomState(1234).randint(6, size=(T, 1)).astype(np.int32)
    try:
        kuka_clib.set_full(True)
        base = kuka_clib.rollout([0], T, actions=actions, aux=True)
        lib.kuka_oracle_set_detail(1)                     # alternating sweep direction of the non-contact rows
        alt = kuka_clib.rollout([0], T, actions=actions, aux=True)
    finally:
        lib.kuka_oracle_set_detail(0)
        kuka_clib.set_full(False)
    assert base["rows"][:, :, 0].sum() == 0               # contact-free
    d = np.abs(alt["q_all"] - base["q_all"]).max()
    assert 1e-4 < d < 0.2
    again = None
    try:
        kuka_clib.set_full(True)
        again = kuka_clib.rollout([0], T, actions=actions, aux=True)
    finally:
        kuka_clib.set_full(False)
    assert np.array_equal(again["q_all"], base["q_all"])   # the bits are off again
