"""Scripted (saturating) policies for the Kuka parity tests: what a trained policy does that a random agent does not — hold an
action until the arm sits in a corner of its workspace box.  Shared by tests/test_kuka_ik_crossing_host.py (kernel source on the
CPU), tests/test_gpu_kuka_ik_crossing.py (the HIP path) and profiles/probes/ik_crossing_extent.py.

Background (round-4 verdict, weak #1): with +x held the arm stretches until joint 3 walks through 0 — the iiwa's elbow singularity —
with every joint at the motors' velocity clamp; Kuka.applyAction's damped-least-squares IK (kuka.py:41-42: jd = 1e-5; kuka.py:144-156)
then has a gain of ~1 / (2 sqrt(jd)) = 158 along the vanishing direction and the closed loop amplifies ANY float64 difference by
~2.4x per step for ~25 steps (5e-12 -> 2e-3 rad).  No two float64 implementations agree beyond such a crossing; both sides flag it
(oracle: kenv.ik_crossed; product: SRLHIP_F_KUKA_IK_CROSSED) and parity is asserted on every env-step BEFORE the flag."""
import itertools

import numpy as np

T_SCRIPT = 1000


def discrete_scripts(T=T_SCRIPT):
    """name -> int32 [T] action sequence.  Actions (kuka_button_gym_env.py:303-308): 0 -x, 1 +x, 2 -y, 3 +y, 4 down, 5 up."""
    out = {}
    for a in range(6):
        out["hold%d" % a] = np.full(T, a, np.int32)
    for sx, sy in itertools.product((0, 1), (2, 3)):
        for first, second in ((sx, sy), (sy, sx)):
            s = np.full(T, 4, np.int32)
            s[:300] = first
            s[300:600] = second
            out["%d_%d_down" % (first, second)] = s
    return out


def continuous_scripts(T=T_SCRIPT):
    """name -> float32 [T][3]: every corner of the saturated action cube."""
    return {"cont%+d%+d%+d" % sgn: np.tile(np.array(sgn, np.float32), (T, 1)) for sgn in itertools.product((-1, 1), repeat=3)}


def joint_scripts(T=T_SCRIPT):
    """name -> float32 [T][7]: joint-space actions at +-1 (no IK on this path: the conditioning flag can never fire)."""
    alt = np.array([1, -1, 1, -1, 1, -1, 1], np.float32)
    return {"joints+": np.ones((T, 7), np.float32), "joints-": -np.ones((T, 7), np.float32), "joints+-": np.tile(alt, (T, 1)),
            "joints-+": np.tile(-alt, (T, 1))}


def batch(scripts, seeds):
    """(names, seeds [n], actions [T][n](...)) for every script x seed."""
    names, ss, cols = [], [], []
    for name, seq in scripts.items():
        for s in seeds:
            names.append("%s/seed%d" % (name, s)); ss.append(s); cols.append(seq)
    return names, np.array(ss, np.int64), np.ascontiguousarray(np.stack(cols, axis=1))


def first_index(mask_t):
    """index of the first True along axis 0 per column, T where there is none"""
    T = mask_t.shape[0]
    return np.where(mask_t.any(axis=0), mask_t.argmax(axis=0), T)


def compare(names, ora, got_q, got_reward, got_done, got_flag=None, tol=1e-7):
    """Parity of one scripted batch (auto_reset off) against the oracle's traces (kuka_clib.rollout(..., ik_trace=True)).
    Asserted: on every env-step of the episode up to (not including) the first step the ORACLE flags as IK-crossed, |dq| <= tol on the
    seven arm joints and reward / done bit for bit; the product's flag trace equals the oracle's unless the oracle's det came within
    1e-6 (relative) of the threshold.  Returned: what happens after the flag (not asserted — see the module docstring)."""
    T, n = ora["done"].shape
    end = np.minimum(first_index(ora["done"] != 0) + 1, T)                 # steps of the (only) episode
    cross = first_index(ora["ik_crossed"] != 0)                            # first flagged step
    stats = {"n": n, "crossed": 0, "post_max_dq": 0.0, "post_flag_mismatch_steps": 0, "post_steps": 0, "pre_max_dq": 0.0, "pre_steps": 0,
             "done_step_moved": 0}
    # the product's full-model kernels are compiled with -fno-honor-nans on the strength of "never produces a NaN": also in this regime
    assert np.isfinite(got_q).all() and np.isfinite(got_reward).all()
    dq = np.abs(ora["q"] - got_q).max(axis=2)                              # [T][n]
    for e in range(n):
        pre = min(int(cross[e]), int(end[e]))
        stats["pre_steps"] += pre
        if pre:
            worst = float(dq[:pre, e].max())
            stats["pre_max_dq"] = max(stats["pre_max_dq"], worst)
            assert worst <= tol, (names[e], worst)
            assert np.array_equal(ora["reward"][:pre, e], got_reward[:pre, e]), names[e]
            assert np.array_equal(ora["done"][:pre, e], got_done[:pre, e]), names[e]
        if cross[e] < end[e]:
            stats["crossed"] += 1
            post = slice(int(cross[e]), int(end[e]))
            stats["post_steps"] += int(end[e] - cross[e])
            stats["post_max_dq"] = max(stats["post_max_dq"], float(dq[post, e].max()))
            stats["post_flag_mismatch_steps"] += int((ora["reward"][post, e] != got_reward[post, e]).sum() + (ora["done"][post, e] != got_done[post, e]).sum())
            g_end = min(first_index(got_done[:, e:e + 1] != 0)[0] + 1, T)
            stats["done_step_moved"] += int(g_end != end[e])
    if got_flag is not None:
        det = ora["ik_det"]
        thr = 3e-9
        near = np.abs(det - thr) < 1e-6 * thr
        assert not near.any(), "an IK det within 1e-6 of the threshold: flags may legitimately differ"
        for e in range(n):
            k = int(end[e])
            assert np.array_equal(ora["ik_crossed"][:k, e], got_flag[:k, e]), names[e]
    return stats
