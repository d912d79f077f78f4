"""CPU: the FULL-model Kuka stepper's own source (csrc/kuka_tree.hpp: 16-lane groups, pointer-jumping kinematics, masked broadcast
sums over the gripper tree, two-bank projected Gauss-Seidel with friction rows) executed on the host — the 16 lanes of a group
are 16 lockstep fibers (csrc/kuka_hostcheck.cpp), the LDS scratch is poisoned with NaN — against the oracle in its full-model
mode (link-frame ABA, dv-space Gauss-Seidel).  Bar: 1e-9 on every joint (north star: 1e-4), discrete flags bit for bit."""
import numpy as np
import pytest

import hostcheck
from oracle import kuka_clib
from srlhip import kuka_model

TOL = 1e-9


@pytest.fixture(autouse=True)
def full_oracle():
    kuka_clib.set_full(True)
    yield
    kuka_clib.set_full(False)


def run(n, T, seed0, variant=0, **kw):
    rs = np.random.RandomState(seed0)
    if kw.get("is_discrete", True):
        actions = rs.randint(6, size=(T, n)).astype(np.int32)
        actions[rs.rand(T, n) < 0.2] = 4                                  # press down: button contacts, friction rows
    else:
        actions = rs.uniform(-1, 1, size=(T, n, 7 if kw.get("action_joints") else 3)).astype(np.float32)
    kuka_clib.set_variant(variant); hostcheck.set_variant(variant)
    try:
        a = kuka_clib.rollout(seed0 + np.arange(n), T, actions=actions, aux=True, **kw)
        b = hostcheck.tree_rollout(seed0 + np.arange(n), T, actions=actions, **kw)
    finally:
        kuka_clib.set_variant(0); hostcheck.set_variant(0)
    assert np.array_equal(a["reward"], b["reward"]) and np.array_equal(a["done"], b["done"])
    assert np.abs(a["reward64"] - b["reward64"]).max() <= TOL
    assert np.abs(a["q"] - b["q"]).max() <= TOL and np.abs(a["gripper"] - b["gripper"]).max() <= TOL
    assert np.abs(a["obs"] - b["obs"]).max() <= 1e-6 and np.abs(a["obs0"] - b["obs0"]).max() <= 1e-6
    assert np.abs(a["final_state"][:, 30:35] - b["final_state"][:, 30:35]).max() <= TOL           # gripper joints
    assert np.array_equal(a["ep_stats"][:, 1:], b["ep_stats"][:, 1:]) and np.abs(a["ep_stats"][:, 0] - b["ep_stats"][:, 0]).max() <= 1e-6   # shaped returns are float sums
    return a


def test_model_tables_of_product_and_oracle_agree():
    assert np.abs(kuka_clib.get_tree_model() - hostcheck.tree_default_model()).max() < 1e-15


def test_discrete_actions_with_contacts_and_friction_rows():
    a = run(6, 900, 5, rng_mode=kuka_clib.RNG_MT19937, random_target=True)
    assert a["rows"][:, :, 0].sum() > 10 and a["rows"][:, :, 1].sum() > 10 and a["done"].sum() >= 4
    # the fingers start open (-+0.3 rad, kuka.py:65-66) and are closed by their 2 / 2.5 N m motors during the 500 settle steps
    assert np.abs(a["q_all"][0, :, [8, 10]]).max() < 0.05


@pytest.mark.parametrize("kw,variant", [
    (dict(rng_mode=kuka_clib.RNG_PHILOX, force_down=False, shape_reward=True), 0),
    (dict(rng_mode=kuka_clib.RNG_MT19937, is_discrete=False), 0),
    (dict(rng_mode=kuka_clib.RNG_MT19937, is_discrete=False, action_joints=True), 0),
    (dict(rng_mode=kuka_clib.RNG_PHILOX, shape_reward=True), 1),
    (dict(rng_mode=kuka_clib.RNG_MT19937, random_target=True, action_repeat=2), 3),
])
def test_options_and_variants(kw, variant):
    run(4, 350, 31, variant=variant, **kw)


def test_joint_limit_rows_and_row_budget_with_a_tightened_model():
    """The general path with joint-LIMIT rows (never reached by a random agent on the real limits): a runtime table whose arm
    limits sit just beyond the settled pose, so that limit rows appear together with contact / friction rows, and a row budget
    of 3 that overflows — oracle and kernel source must drop the same candidates."""
    t = kuka_clib.get_tree_model().copy()
    J = 1 + 33 * np.arange(12)
    q_settled = np.array([0.0, 0.6, 0.0, -0.86, 0.0, 1.68, 0.0])              # roughly where the 500 settle steps leave the arm
    t[J[:7] + 16] = q_settled - 0.12                                          # lower limits
    t[J[:7] + 17] = q_settled + 0.12                                          # upper limits
    t[kuka_model.TREE_MAX_GENERIC_ROWS] = 3.0                                                               # max_generic_rows
    try:
        kuka_clib.set_tree_model(t); hostcheck.tree_set_model(t)
        a = run(4, 600, 91, rng_mode=kuka_clib.RNG_PHILOX, random_target=True)
        lim = a["rows"][:, :, 1] // 1000
        assert (lim > 0).sum() > 50 and ((lim > 0) & (a["rows"][:, :, 0] > 0)).sum() >= 0
    finally:
        hostcheck.tree_set_model(None)
        kuka_clib.set_full(True)


def two_button_actions_full(n, T, seed=5):
    """Kuka2Button with the full model: random walks, env 0 scripted to press button 1 (y = +0.125) five times, rise, move over and
    press button 2 (open loop, found with the oracle: the default-damping IK of this env tracks its target loosely), env 1 idles
    into the 1500-step limit, env 2 goes for the second button first."""
    actions = np.random.RandomState(seed).randint(6, size=(T, n)).astype(np.int32)
    script = [3] * 4 + [0] * 4 + [4] * 17 + [-1] * 400 + [5] * 8 + [-1] * 120 + [2] * 8 + [-1] * 280 + [4] * 20
    actions[:len(script), 0] = script
    actions[len(script):, 0] = -1
    actions[:, 1] = -1
    script2 = [2] * 4 + [4] * 50
    actions[:len(script2), 2] = script2
    return actions


def test_two_button_variant():
    """Kuka2ButtonGymEnv on the tree kernel's source (NB = 2: the second button's motor / stop rows on the same three lanes, its
    cap and base as contact shapes, rows that act on glider 1 or 2, goal switching) against the full-model oracle."""
    n, T = 4, 1600
    actions = two_button_actions_full(n, T)
    try:
        kuka_clib.set_variant(2); hostcheck.set_variant(2)
        for kw in (dict(force_down=False, max_distance=2.0), dict(force_down=False, max_distance=2.0, shape_reward=True, random_target=True)):
            a = kuka_clib.rollout(60 + np.arange(n), T, actions=actions, aux=True, **kw)
            b = hostcheck.tree_rollout(60 + np.arange(n), T, actions=actions, **kw)
            assert np.array_equal(a["reward"], b["reward"]) and np.array_equal(a["done"], b["done"])
            assert np.abs(a["reward64"] - b["reward64"]).max() <= TOL
            assert np.abs(a["q"] - b["q"]).max() <= TOL and np.abs(a["gripper"] - b["gripper"]).max() <= TOL
            assert np.abs(a["final_state"][:, 30:35] - b["final_state"][:, 30:35]).max() <= TOL           # gripper joints
            assert np.array_equal(a["final_state"][:, 26:28], b["final_state"][:, 26:28])                 # goal_id, n_contacts[1]
            assert np.abs(a["final_state"][:, 24:26] - b["final_state"][:, 24:26]).max() <= TOL           # second glider
            assert np.array_equal(a["final_state"][:, 28:30], b["final_state"][:, 28:30])                 # second button position
            assert np.array_equal(a["ep_stats"][:, 1:], b["ep_stats"][:, 1:])
            if not kw.get("random_target"):
                first = np.argmax(a["done"][:, 0])
                assert a["done"][first, 0] and 900 < first < 1400 and a["reward"][first, 0] == 1.0     # both buttons pressed in order
                assert a["reward"][:first, 0].sum() == 4 and (a["reward"][:first, 0] != 0).sum() == 4   # sparse: last button only
                assert a["rows"][:, :, 0].sum() > 100 and a["rows"][:, 2, 0].sum() > 10                 # contact rows, also on button 2 first
        assert a["ep_stats"][:, 1].max() == 1501
    finally:
        kuka_clib.set_variant(0); hostcheck.set_variant(0)


def test_two_button_variant_with_joint_limit_rows():
    """NB = 2 on the general path's LDS loop (joint-limit rows in the wavefront): limits of joints 3 and 5 tightened to 0.3 rad
    around the settled pose, row budget 3 — limit rows, contact rows on either button and the second button's rows interleave."""
    t = kuka_clib.get_tree_model().copy()
    J = 1 + 33 * np.arange(12)
    q_settled = np.array([0.0, 0.6, 0.0, -0.86, 0.0, 1.68, 0.0])
    jj = np.array([3, 5])
    t[J[jj] + 16] = q_settled[jj] - 0.3
    t[J[jj] + 17] = q_settled[jj] + 0.3
    t[kuka_model.TREE_MAX_GENERIC_ROWS] = 3.0
    n, T = 6, 700
    rs = np.random.RandomState(9)
    actions = rs.randint(6, size=(T, n)).astype(np.int32)
    actions[rs.rand(T, n) < 0.35] = 4
    kw = dict(force_down=False, max_distance=2.0, random_target=True)
    try:
        kuka_clib.set_tree_model(t); hostcheck.tree_set_model(t)
        kuka_clib.set_variant(2); hostcheck.set_variant(2)
        a = kuka_clib.rollout(20 + np.arange(n), T, actions=actions, aux=True, **kw)
        b = hostcheck.tree_rollout(20 + np.arange(n), T, actions=actions, **kw)
    finally:
        kuka_clib.set_variant(0); hostcheck.set_variant(0)
        hostcheck.tree_set_model(None); kuka_clib.set_full(True)
    lim, normals = a["rows"][:, :, 1] // 1000, a["rows"][:, :, 0]
    assert (lim > 0).mean() > 0.3 and ((lim > 0) & (normals > 0)).sum() > 50
    assert np.array_equal(a["reward"], b["reward"]) and np.array_equal(a["done"], b["done"])
    assert np.abs(a["q"] - b["q"]).max() <= TOL and np.abs(a["final_state"][:, 24:26] - b["final_state"][:, 24:26]).max() <= TOL
    assert np.array_equal(a["final_state"][:, 26:28], b["final_state"][:, 26:28])


# ---- solver details as data (round 4): every bit / scalar of srlhip_kuka_tree_model's solver section, kernel source vs oracle
def detail_table(detail=0, contact_erp=None, limit_erp=None, linear_slop=None, tighten=False, budget=None):
    t = kuka_clib.get_tree_model().copy()
    t[kuka_model.TREE_SOLVER_DETAIL] = detail
    if contact_erp is not None:
        t[kuka_model.TREE_CONTACT_ERP] = contact_erp
    if limit_erp is not None:
        t[kuka_model.TREE_LIMIT_ERP] = limit_erp
    if linear_slop is not None:
        t[kuka_model.TREE_LINEAR_SLOP] = linear_slop
    if tighten:                                                   # arm limits just beyond the settled pose: limit rows appear
        J = kuka_model.TREE_JOINT0 + kuka_model.TREE_JOINT_STRIDE * np.arange(7)
        q_settled = np.array([0.0, 0.6, 0.0, -0.86, 0.0, 1.68, 0.0])
        t[J + kuka_model.TREE_LOWER] = q_settled - 0.12
        t[J + kuka_model.TREE_UPPER] = q_settled + 0.12
    if budget is not None:
        t[kuka_model.TREE_MAX_GENERIC_ROWS] = budget
    return t


_BASE = {}


@pytest.mark.parametrize("detail", [1, 2, 4, 7])
def test_solver_detail_bits_free_and_contact_steps(detail):
    """Alternating sweep (1), body-creation order (2), second friction direction (4) and all three: the kernel source integrates
    the same trajectory as the oracle with the same bits, and a DIFFERENT one from the default order (the bits are not no-ops)."""
    if "q" not in _BASE:
        _BASE["q"] = run(6, 900, 5, rng_mode=kuka_clib.RNG_MT19937, random_target=True)["q"]
    base = _BASE
    t = detail_table(detail)
    try:
        kuka_clib.set_tree_model(t); hostcheck.tree_set_model(t)
        a = run(6, 900, 5, rng_mode=kuka_clib.RNG_MT19937, random_target=True)
    finally:
        hostcheck.tree_set_model(None); kuka_clib.set_full(True)
    normals, fric = a["rows"][:, :, 0].sum(), (a["rows"][:, :, 1] % 1000).sum()
    assert normals > 10 and fric == (2 if detail & 4 else 1) * normals      # contact steps were part of it; friction rows per contact
    assert np.abs(a["q"] - base["q"]).max() > 1e-6


@pytest.mark.parametrize("detail", [0, 3, 7])
def test_solver_detail_bits_with_joint_limit_rows(detail):
    """The general path's LDS loop under the detail bits: limit rows (swept inside the non-contact segment, backwards on even
    iterations with bit 0) together with contact and friction rows, row budget 3 overflowing (2 with two friction directions...)."""
    t = detail_table(detail, tighten=True, budget=3)
    try:
        kuka_clib.set_tree_model(t); hostcheck.tree_set_model(t)
        a = run(4, 500, 91, rng_mode=kuka_clib.RNG_PHILOX, random_target=True)
    finally:
        hostcheck.tree_set_model(None); kuka_clib.set_full(True)
    assert ((a["rows"][:, :, 1] // 1000) > 0).sum() > 50


def test_erp_and_slop_are_data():
    t = detail_table(0, contact_erp=0.08, limit_erp=0.1, linear_slop=1e-5, tighten=True)
    base = run(4, 400, 17, rng_mode=kuka_clib.RNG_PHILOX, random_target=True)
    try:
        kuka_clib.set_tree_model(t); hostcheck.tree_set_model(t)
        a = run(4, 400, 17, rng_mode=kuka_clib.RNG_PHILOX, random_target=True)
    finally:
        hostcheck.tree_set_model(None); kuka_clib.set_full(True)
    assert np.abs(a["q"] - base["q"]).max() > 1e-6


def test_two_button_variant_under_detail_bits():
    n, T = 4, 900
    actions = two_button_actions_full(n, T)
    t = detail_table(7)
    kw = dict(force_down=False, max_distance=2.0)
    try:
        kuka_clib.set_tree_model(t); hostcheck.tree_set_model(t)
        kuka_clib.set_variant(2); hostcheck.set_variant(2)
        a = kuka_clib.rollout(60 + np.arange(n), T, actions=actions, aux=True, **kw)
        b = hostcheck.tree_rollout(60 + np.arange(n), T, actions=actions, **kw)
    finally:
        kuka_clib.set_variant(0); hostcheck.set_variant(0)
        hostcheck.tree_set_model(None); kuka_clib.set_full(True)
    assert a["rows"][:, :, 0].sum() > 20
    assert np.array_equal(a["reward"], b["reward"]) and np.array_equal(a["done"], b["done"])
    assert np.abs(a["q"] - b["q"]).max() <= TOL and np.abs(a["final_state"][:, 24:26] - b["final_state"][:, 24:26]).max() <= TOL
    assert np.array_equal(a["final_state"][:, 26:28], b["final_state"][:, 26:28])


# ---- KukaRandButtonGymEnv free bodies (round 4)
def test_rand_button_free_bodies():
    """The ten distractors and the kicked ball as free bodies (kuka_rand_button_gym_env.py:59-71,111-125): arm-sphere <-> body contact
    rows (normal + friction, acting on arm AND body), the bodies' table-contact and table-friction rows, the seeded kick — kernel
    source vs oracle, default solver details and all bits; object poses compared at the end of the run."""
    import ctypes
    n, T, seed0 = 6, 500, 3
    rs = np.random.RandomState(seed0)
    actions = rs.randint(6, size=(T, n)).astype(np.int32)
    actions[rs.rand(T, n) < 0.5] = 4
    assert kuka_clib.rb_drop_check() < 1e-12                     # the literal drop of the reference's reset ends in the rest state reset() uses
    for detail in (0, 7):
        t = detail_table(detail)
        hb = np.zeros((n, 11, 7))
        try:
            kuka_clib.set_tree_model(t); hostcheck.tree_set_model(t)
            kuka_clib.set_variant(3); hostcheck.set_variant(3)
            ob = kuka_clib.body_trace(n)
            hostcheck.lib().hostcheck_kuka_tree_set_body_trace(hb.ctypes.data_as(ctypes.c_void_p))
            kw = dict(rng_mode=kuka_clib.RNG_MT19937, random_target=True)
            a = kuka_clib.rollout(seed0 + np.arange(n), T, actions=actions, aux=True, **kw)
            b = hostcheck.tree_rollout(seed0 + np.arange(n), T, actions=actions, **kw)
        finally:
            kuka_clib.body_trace_off(); hostcheck.lib().hostcheck_kuka_tree_set_body_trace(None)
            kuka_clib.set_variant(0); hostcheck.set_variant(0)
            hostcheck.tree_set_model(None); kuka_clib.set_full(True)
        assert a["rows"][:, :, 2].sum() > 50                     # arm <-> body contact rows
        assert np.array_equal(a["reward"], b["reward"]) and np.array_equal(a["done"], b["done"])
        assert np.abs(a["q"] - b["q"]).max() <= TOL and np.abs(a["gripper"] - b["gripper"]).max() <= TOL
        assert np.abs(ob - hb).max() <= TOL
        assert (ob[:, 10, 0] > 0.25 + 1e-3).any() and (ob[:, 10, 1] > -0.2 + 1e-3).any()      # kicked balls have moved towards +x, +y (they stop within ~40 steps)


# ---- the two-wavefronts-per-SIMD variant's code path (OCC = 1, round 4)
@pytest.mark.parametrize("detail,tighten", [(0, False), (0, True), (3, True)])
def test_occ_variant_shared_work_area_and_recomputed_candidates(detail, tighten):
    """kuka_tree_occ.hip's kernels run tphysics_step / general_path with OCC = 1: one general-path work area per wavefront taken in
    turns, the own row of M^-1 and the spatial axes parked per env, sphere / limit candidates recomputed inside the turn.  Same
    trajectories as the oracle (and therefore as the default variant): contact + friction steps, joint-limit rows, detail bits."""
    t = detail_table(detail, tighten=tighten, budget=3 if tighten else None)
    try:
        kuka_clib.set_tree_model(t); hostcheck.tree_set_model(t)
        hostcheck.lib().hostcheck_kuka_tree_set_occ(1)
        a = run(6, 900, 5, rng_mode=kuka_clib.RNG_MT19937, random_target=True) if not tighten else run(4, 600, 91, rng_mode=kuka_clib.RNG_PHILOX, random_target=True)
    finally:
        hostcheck.lib().hostcheck_kuka_tree_set_occ(0)
        hostcheck.tree_set_model(None); kuka_clib.set_full(True)
    if tighten:
        assert ((a["rows"][:, :, 1] // 1000) > 0).sum() > 50
    else:
        assert a["rows"][:, :, 0].sum() > 10
