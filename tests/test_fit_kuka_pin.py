"""CPU: tests/golden/fit_kuka_pin.py recovers hidden solver details from a fixture.  The real fixture needs PyBullet (absent here);
these synthetic ones are written by the oracle itself with details it is then not told — the search must find them, meet the
1e-4 bar only with the right combination, and tell the order bits apart on contact-free steps alone."""
import os
import sys

import numpy as np
import pytest

from oracle import kuka_clib
from srlhip import kuka_model

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import fit_kuka_pin  # noqa: E402


def synthetic_fixture(detail, scalars, seeds=(0, 1), T=600):
    kuka_clib.set_full(True)
    t = kuka_clib.get_tree_model().copy()
    t[kuka_model.TREE_SOLVER_DETAIL] = detail
    t[kuka_model.TREE_CONTACT_ERP], t[kuka_model.TREE_LIMIT_ERP], t[kuka_model.TREE_LINEAR_SLOP] = scalars
    rec = {k: [] for k in ("seed", "action", "q", "q14", "reward", "done")}
    try:
        kuka_clib.set_tree_model(t)
        for seed in seeds:
            rs = np.random.RandomState(100 + seed)
            actions = np.full(T, 4, np.int32)                                # straight down: ~550 steps to the button, then contact + friction rows
            actions[:20] = rs.randint(4, size=20)                            # a little x / y wandering first
            out = kuka_clib.rollout([seed], T, actions=actions[:, None], aux=True)
            q14 = np.zeros((T, 14))
            q14[:, [0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 11, 13]] = out["q_all"][:, 0]
            rec["seed"] += [seed] * T; rec["action"] += list(actions); rec["q"] += list(out["q"][:, 0]); rec["q14"] += list(q14)
            rec["reward"] += list(out["reward64"][:, 0]); rec["done"] += list(out["done"][:, 0].astype(int))
            assert out["rows"][:, 0, 0].sum() > 3                            # the record contains contact steps
    finally:
        kuka_clib.set_full(False)
    fx = {k: np.asarray(v) for k, v in rec.items()}
    base = t.copy()                                                          # what a fixture carries: the model, details at their defaults
    base[kuka_model.TREE_SOLVER_DETAIL], base[kuka_model.TREE_CONTACT_ERP], base[kuka_model.TREE_LIMIT_ERP], base[kuka_model.TREE_LINEAR_SLOP] = 0, 0.2, 0.2, 0.0
    fx["tree_model_table"] = base
    return fx


@pytest.mark.parametrize("detail,scalars", [(3, (0.2, 0.2, 0.0)), (6, (0.08, 0.2, 1e-5))])
def test_fit_recovers_hidden_details(detail, scalars):
    fx = synthetic_fixture(detail, scalars)
    results, order_rank = fit_kuka_pin.fit(fx, scalar_grid=((0.2, 0.2, 0.0), (0.08, 0.2, 1e-5)), verbose=False)
    best = results[0]
    assert best["detail"] == detail and (best["contact_erp"], best["limit_erp"], best["linear_slop"]) == scalars
    assert best["flags_ok"] and best["err_all"] < 1e-9
    # the contact-free starts alone identify the two ORDER bits (stage 1 of the search)
    assert order_rank[0][1] == detail & 3 and order_rank[0][0] < 1e-9 and order_rank[1][0] > 1e-6
    # and nothing else meets the bar on the whole record
    wrong = [r for r in results[1:] if r["detail"] & 3 != detail & 3]
    assert all(r["err_all"] > fit_kuka_pin.TOL or not r["flags_ok"] for r in wrong)


def test_old_fixture_tables_are_padded():
    fx = {"tree_model_table": np.zeros(kuka_model.TREE_MODEL_DOUBLES - 4)}
    t = fit_kuka_pin.table_of(fx)
    assert t.shape == (kuka_model.TREE_MODEL_DOUBLES,) and list(t[-4:]) == [0.0, 0.2, 0.2, 0.0]


def test_recorder_helpers_of_the_pybullet_recipe():
    """make_kuka_pybullet_golden.py's round-4 recorders (contact friction directions, engine parameters, the settle-step probe)
    against a minimal fake pybullet: fixed-size NaN-padded arrays, both getContactPoints tuple lengths, the wrapper restored."""
    import json
    import types
    import make_kuka_pybullet_golden as mk
    calls = {"n": 0}
    long_pt = tuple([0] * 7) + ((0.0, 0.0, 1.0), -1e-4, 3.5, 0.2, (1.0, 0.0, 0.0), -0.1, (0.0, 1.0, 0.0))
    short_pt = tuple([0] * 7) + ((0.0, 1.0, 0.0), 2e-4, 0.0)
    p = types.SimpleNamespace(
        getContactPoints=lambda a, b, link=None: [long_pt, short_pt],
        getPhysicsEngineParameters=lambda: {"erp": 0.2, "contactERP": 0.08, "numSolverIterations": 150, "weird": (1, 2)},
        getJointState=lambda uid, j: (0.1 * j + calls["n"], -0.01 * j, (0,) * 6, 0.0),
        stepSimulation=lambda: calls.__setitem__("n", calls["n"] + 1))
    rec, n = mk.contact_record(p, 2, 3)
    assert n == 2 and rec["normal"].shape == (mk.MAX_CONTACTS, 3) and rec["fric2"][0] == -0.1 and np.isnan(rec["fric2"][1])
    assert list(rec["fric_dir2"][0]) == [0.0, 1.0, 0.0] and rec["distance"][1] == 2e-4 and np.isnan(rec["distance"][2])
    d = json.loads(mk.engine_parameters(p))
    assert d["contactERP"] == 0.08 and d["numSolverIterations"] == 150 and d["weird"] == "(1, 2)"
    real = p.stepSimulation
    with mk.SettleProbe(p, lambda: 3) as probe:
        for _ in range(mk.SETTLE_PROBE + 5):
            p.stepSimulation()
    assert p.stepSimulation is real and np.array(probe.q).shape == (mk.SETTLE_PROBE, 14) and probe.q[1][0] == 2.0 and probe.qd[0][3] == -0.03
