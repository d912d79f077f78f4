"""CPU: the raster oracle's full-model gripper (round 4: the HIP rasteriser draws the gripper from joints 7, 8, 10, 11, 13 on
full-model handles; oracle/raster_oracle.c `kind | 16` is its checker).  Closed fingers, opened fingers and the lumped drawing give
different frames; the finger capsules follow the joint angles; without the full-model table the flag is refused."""
import numpy as np
import pytest

from oracle import kuka_clib, raster_clib

Q = np.array([[0.0, 0.6, 0.0, -0.86, 0.0, 1.68, 0.0, 0.0, 0.5, 0.0]])          # q7 near the settled pose, bq, bx, by


def test_fingers_follow_their_joints():
    kuka_clib.set_full(True)
    try:
        closed = raster_clib.render(4, Q, 128, 128, gripper_q=np.zeros((1, 5)))
        opened = raster_clib.render(4, Q, 128, 128, gripper_q=np.array([[0.0, -0.6, 0.0, 0.6, 0.0]]))
        turned = raster_clib.render(4, Q, 128, 128, gripper_q=np.array([[1.2, 0.0, 0.0, 0.0, 0.0]]))
        lumped = raster_clib.render(4, Q, 128, 128)
    finally:
        kuka_clib.set_full(False)
    for a, b in ((closed, opened), (closed, turned), (closed, lumped)):
        d = (a != b).any(axis=-1).sum()
        assert 10 < d < 2000, d                        # a gripper-sized difference, not a different scene
    # lumped drawing does not depend on the oracle's model mode
    assert np.array_equal(lumped, raster_clib.render(4, Q, 128, 128))


def test_finger_flag_needs_the_full_model():
    kuka_clib.set_full(False)
    with pytest.raises(AssertionError):
        raster_clib.render(4, Q, 64, 64, gripper_q=np.zeros((1, 5)))
