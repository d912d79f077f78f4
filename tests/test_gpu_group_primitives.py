"""-m gpu: the cross-lane primitives of the lane-group Kuka kernel (csrc/kuka_group.hpp) on the device — DPP row
broadcasts / shifts, row votes, the fused projected-Gauss-Seidel row instructions (v_add_f64 clamp + v_fma_f64 +
v_fmac_f64_dpp), the lane-parallel Gauss-Jordan and the prefix-composed forward kinematics — against their definitions
(the ones the CPU-side fiber emulation of the same source implements)."""
import ctypes

import numpy as np
import pytest

from oracle import kuka_clib
from srlhip import _lib

pytestmark = pytest.mark.gpu


def probe(q7):
    lib = _lib.load()
    q7 = np.ascontiguousarray(q7, dtype=np.float64)
    out = np.zeros((40, 64))
    rc = lib.srlhip_selftest_group_primitives(0, q7.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p), out.size)
    assert rc == 0
    return out


def test_group_primitives_match_their_definitions():
    q7 = np.array([0.3, -0.7, 0.2, -1.3, 0.4, 1.1, -0.5])
    o = probe(q7)
    t = np.arange(64); l = t % 16; base = t - l
    x = 1.5 * t + 0.25
    k = iter(range(40))
    assert np.array_equal(o[next(k)], l)
    assert np.array_equal(o[next(k)], x[base + 3]) and np.array_equal(o[next(k)], x[base + 15])
    for d in (1, 2, 4):
        assert np.array_equal(o[next(k)], np.where(l >= d, x[np.maximum(t - d, 0)], -float(d)))
    votes = np.array([sum(1 << i for i in range(16) if (b + i) % 3 == 0) for b in base])
    assert np.array_equal(o[next(k)], votes)
    assert np.array_equal(o[next(k)], (base == 32).astype(float)) and np.array_equal(o[next(k)], np.ones(64))
    assert np.array_equal(o[next(k)], t + 2.0 * x[base + 5])                       # fmac_bcast<5>
    # pgs_row<2>: t = clamp01(cs + acc); acc -= ep * acc; acc += n * bcast<2>(t)
    acc, cs, ep = 0.125 * t, 0.125 * l - 0.25, (l == 4).astype(float)
    tt = np.clip(cs + acc, 0.0, 1.0)
    assert np.array_equal(o[next(k)], (acc - ep * acc) + 0.5 * tt[base + 2])
    assert np.array_equal(o[next(k)], tt)
    acc, cs, ep = 0.125 * t, 0.25 * l - 0.5, (l == 0).astype(float)
    tt = np.clip(cs + acc, 0.0, 1.0)
    want = (acc - ep * acc) + 0.5 * tt[base + 1]
    want = want + np.where(l >= 8, 0.25, 0.0) * tt[base + 9]
    assert np.array_equal(o[next(k)], want) and np.array_equal(o[next(k)], tt)
    assert np.abs(o[next(k)] * (x + 1.0) - 1.0).max() < 4e-16                      # rcp
    le = (np.arange(7)[None, :] <= l[:, None]).astype(float)
    ge = ((np.arange(7)[None, :] >= l[:, None]) & (l[:, None] < 7)).astype(float)
    xs = x[base[:, None] + np.arange(7)[None, :]]
    assert np.allclose(o[next(k)], (le * xs).sum(1), rtol=1e-15, atol=0)
    assert np.allclose(o[next(k)], 3.0 + (ge * xs).sum(1), rtol=1e-15, atol=0)
    # Gauss-Jordan: solve and inverse of the 7x7 SPD test matrix, row i on lane i
    A = np.array([[4.0 + i if i == j else 1.0 / (1.0 + i + j) for j in range(7)] for i in range(7)])
    sol = np.linalg.solve(A, 1.0 + np.arange(7))
    got = o[next(k)]
    inv = np.stack([o[next(k)] for _ in range(7)], axis=1)                          # [lane][col]
    for b in (0, 16, 32, 48):
        assert np.abs(got[b:b + 7] - sol).max() < 1e-14 and np.all(got[b + 7:b + 16] == 0)
        assert np.abs(inv[b:b + 7] - np.linalg.inv(A)).max() < 1e-14
    # forward kinematics by prefix composition vs the oracle's chain walk; gripper point
    R = np.stack([o[next(k)] for _ in range(9)], axis=1)                            # [lane][9], columns x y z
    p = np.stack([o[next(k)] for _ in range(3)], axis=1)
    grip = np.stack([o[next(k)] for _ in range(3)], axis=1)
    Ro, po = kuka_clib.fk(q7)                                                       # Ro[i][row][col]
    for b in (0, 16, 32, 48):
        for i in range(7):
            assert np.abs(R[b + i].reshape(3, 3).T - Ro[i]).max() < 1e-14, i
            assert np.abs(p[b + i] - po[i]).max() < 1e-14, i
    want_grip = po[6] + Ro[6] @ np.array([0, 0.024, 0.10])
    assert np.abs(grip - want_grip).max() < 1e-14
