"""ctypes front end of oracle/raster_oracle.c.  TEST INFRASTRUCTURE ONLY."""
import ctypes

import numpy as np

from . import clib


def render(kind, state, h=64, w=64, multi_view=False, gripper_q=None):
    """gripper_q [n][5] (Kuka kinds; q of joints 7, 8, 10, 11, 13 as srlhip_get_state(KUKA_GRIPPER_Q) returns them, transposed): draw
    the gripper from its own joints through the full model's link frames (the oracle must be in full-model mode) — what the HIP
    rasteriser does on full-model handles; None: the gripper welded to link 7 (lumped handles).
    kind 0..3 mobile family (state [n][6]: x y tx ty t2x t2y), 4 kuka (state [n][10]: q7 bq bx by),
    6 kuka with two buttons (state [n][13]: q7 bq bx by b2q b2x b2y), 7 kuka with the RandButton distractors
    (state [n][40]: q7 bq bx by + (x, y, present) x 10), 8 the same with the free bodies' state (state [n][106]: + (x y z vx vy vz) x 11)."""
    state = np.ascontiguousarray(state, dtype=np.float64)
    n = len(state)
    assert state.shape == (n, {4: 10, 6: 13, 7: 40, 8: 106}.get(kind, 6))
    if gripper_q is not None:
        gq = np.asarray(gripper_q, dtype=np.float64)
        assert kind >= 4 and gq.shape == (n, 5)
        state = np.ascontiguousarray(np.concatenate([state, gq], axis=1))
        kind |= 16
    img = np.zeros((n, h, w, 6 if multi_view else 3), np.uint8)
    rc = clib.lib().raster_oracle_render(int(kind), n, int(h), int(w), int(bool(multi_view)),
                                         state.ctypes.data_as(ctypes.c_void_p), img.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0, "raster_oracle_render failed (gripper_q needs the oracle in full-model mode: kuka_clib.set_full(True))"
    return img
