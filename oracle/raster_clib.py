"""ctypes front end of oracle/raster_oracle.c.  TEST INFRASTRUCTURE ONLY."""
import ctypes

import numpy as np

from . import clib


def render(kind, state, h=64, w=64, multi_view=False):
    """kind 0..3 mobile family (state [n][6]: x y tx ty t2x t2y), 4 kuka (state [n][10]: q7 bq bx by),
    6 kuka with two buttons (state [n][13]: q7 bq bx by b2q b2x b2y), 7 kuka with the RandButton distractors
    (state [n][40]: q7 bq bx by + (x, y, present) x 10), 8 the same with the free bodies' state (state [n][106]: + (x y z vx vy vz) x 11)."""
    state = np.ascontiguousarray(state, dtype=np.float64)
    n = len(state)
    assert state.shape == (n, {4: 10, 6: 13, 7: 40, 8: 106}.get(kind, 6))
    img = np.zeros((n, h, w, 6 if multi_view else 3), np.uint8)
    clib.lib().raster_oracle_render(int(kind), n, int(h), int(w), int(bool(multi_view)),
                                    state.ctypes.data_as(ctypes.c_void_p), img.ctypes.data_as(ctypes.c_void_p))
    return img
