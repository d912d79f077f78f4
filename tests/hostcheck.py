"""Loader for csrc/build/libsrlhip_hostcheck.so — the HIP stepper's own physics
source (kuka_core.hpp / kuka_env.hpp) compiled for the host.  TESTS ONLY."""
import ctypes
import os
import subprocess
import types

from oracle import kuka_clib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "robotics-rl-srl_amd", "csrc")
LIB = os.path.join(CSRC, "build", "libsrlhip_hostcheck.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        subprocess.check_call(["make", "-s", "-C", CSRC, "hostcheck"])
        _lib = ctypes.CDLL(LIB)
        _lib.hostcheck_kuka_rollout.argtypes = kuka_clib._lib().kuka_oracle_rollout.argtypes
    return _lib


def set_moving(flag):
    lib().hostcheck_kuka_set_moving(int(bool(flag)))


def set_variant(variant):
    lib().hostcheck_kuka_set_variant(int(variant))


def rollout(seeds, T, actions=None, **kw):
    """Same arguments / outputs as oracle.kuka_clib.rollout, computed by the kernel source on the host."""
    fake = types.SimpleNamespace(kuka_oracle_rollout=lib().hostcheck_kuka_rollout)
    real = kuka_clib._lib
    kuka_clib._lib = lambda: fake
    try:
        return kuka_clib.rollout(seeds, T, actions=actions, **kw)
    finally:
        kuka_clib._lib = real


def group_rollout(seeds, T, actions=None, **kw):
    """Same, computed by the lane-GROUP stepper (csrc/kuka_group.hpp) with its 16 lanes emulated as lockstep fibers."""
    l = lib()
    l.hostcheck_kuka_group_rollout.argtypes = l.hostcheck_kuka_rollout.argtypes
    fake = types.SimpleNamespace(kuka_oracle_rollout=l.hostcheck_kuka_group_rollout)
    real = kuka_clib._lib
    kuka_clib._lib = lambda: fake
    try:
        return kuka_clib.rollout(seeds, T, actions=actions, **kw)
    finally:
        kuka_clib._lib = real


def set_model(table):
    """Runtime model table (138 doubles or a dict, see oracle.kuka_clib.MODEL_FIELDS) for group_rollout(); None -> baked model."""
    import numpy as np
    if table is None:
        lib().hostcheck_kuka_set_model(None)
        return
    t = np.ascontiguousarray(kuka_clib.model_to_table(table) if isinstance(table, dict) else table, dtype=np.float64)
    assert t.shape == (kuka_clib.MODEL_DOUBLES,)
    lib().hostcheck_kuka_set_model(t.ctypes.data_as(ctypes.c_void_p))


def default_model():
    import numpy as np
    t = np.zeros(kuka_clib.MODEL_DOUBLES)
    lib().hostcheck_kuka_default_model(t.ctypes.data_as(ctypes.c_void_p))
    return kuka_clib.model_to_dict(t)


def tree_rollout(seeds, T, actions=None, ik_trace=False, **kw):
    """Same, computed by the FULL-model lane-group stepper (csrc/kuka_tree.hpp: 12-DoF gripper tree, contact spheres per link,
    friction rows) under the fiber harness; compare with oracle.kuka_clib.rollout after kuka_clib.set_full(True).
    ik_trace: also "ik_crossed" [T][n] (the kernel source's sticky IK conditioning bit after each step) and "ik_final" [n][2]
    (bit, flagged env-steps) unpacked from Env::ikx — same layout as the oracle's."""
    import numpy as np
    l = lib()
    if ik_trace:
        n = len(seeds)
        flag, fin = np.zeros((T, n), np.uint8), np.zeros(n, np.int32)
        l.hostcheck_kuka_tree_set_ik_trace(flag.ctypes.data_as(ctypes.c_void_p), fin.ctypes.data_as(ctypes.c_void_p))
        try:
            out = tree_rollout(seeds, T, actions=actions, **kw)
        finally:
            l.hostcheck_kuka_tree_set_ik_trace(None, None)
        out["ik_crossed"], out["ik_final"] = flag, np.stack([fin & 1, fin >> 1], axis=1)
        return out
    l.hostcheck_kuka_tree_rollout.argtypes = l.hostcheck_kuka_rollout.argtypes
    fake = types.SimpleNamespace(kuka_oracle_rollout=l.hostcheck_kuka_tree_rollout)
    real = kuka_clib._lib
    kuka_clib._lib = lambda: fake
    try:
        return kuka_clib.rollout(seeds, T, actions=actions, **kw)
    finally:
        kuka_clib._lib = real


def tree_default_model():
    import numpy as np
    t = np.zeros(510)
    lib().hostcheck_kuka_tree_default_model(t.ctypes.data_as(ctypes.c_void_p))
    return t


def tree_set_model(table):
    """Runtime full-model table (510 doubles) for tree_rollout(); None -> the baked model."""
    import numpy as np
    if table is None:
        lib().hostcheck_kuka_tree_set_model(None)
        return
    t = np.ascontiguousarray(table, dtype=np.float64)
    assert t.shape == (510,)
    lib().hostcheck_kuka_tree_set_model(t.ctypes.data_as(ctypes.c_void_p))
