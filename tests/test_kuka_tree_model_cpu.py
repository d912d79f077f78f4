"""CPU: the full Kuka model as data.  (1) The product's baked srlhip_kuka_tree_model equals the oracle's table; (2) the dict / table
round trip; (3) srlhip.kuka_model.tree_from_pybullet — what tests/golden/make_kuka_pybullet_golden.py runs against the real
PyBullet — reproduces the baked table when it is fed a fake pybullet module that serves the same kuka_with_gripper2 description
(link frames at q = 0, inertial frames, the two fixed finger_base joints): the extraction and the fixed-link merge are checked
without PyBullet."""
import numpy as np

from oracle import kuka_clib
from srlhip import _lib, kuka_model


def test_product_and_oracle_tables_agree_and_round_trip():
    t = _lib.kuka_tree_default_model()
    kuka_clib.set_full(True)
    try:
        o = kuka_clib.get_tree_model()
    finally:
        kuka_clib.set_full(False)
    assert t.shape == (510,) and np.abs(t - o).max() < 1e-15
    m = kuka_model.tree_to_dict(t)
    assert m["nd"] == 12 and m["nsphere"] == 16 and m["ee_link"] == 6 and m["grip_link"] == 8
    assert [int(j["joint_index"]) for j in m["joints"]] == [0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 11, 13]
    assert [int(j["parent"]) for j in m["joints"]] == [-1, 0, 1, 2, 3, 4, 5, 6, 7, 8, 7, 10]
    assert np.array_equal(kuka_model.tree_to_table(m), t)
    # motors of kuka.py:33-38,167-187: arm 200 N m / 0.35 rad/s / gain 0.3, joint 7 200, fingers 2 and 2.5, tips 2, default gain 0.1
    assert [j["max_force"] for j in m["joints"]] == [200.0] * 8 + [2.0, 2.0, 2.5, 2.0]
    assert [j["kp"] for j in m["joints"]] == [0.3] * 7 + [0.1] * 5


class FakeBullet:
    """Serves the recalled kuka_with_gripper2.sdf the way pybullet's introspection calls do (14 joints, 9 and 12 fixed)."""
    JOINT_FIXED = 4

    def __init__(self):
        def rpy(r):
            return kuka_model._rpy_matrix(r)
        arm = kuka_model.default()
        T = np.eye(4)
        self.frames, self.parent, self.axis, self.lim, self.damp, self.dyn, self.fixed = [], [], [], [], [], [], []
        for i in range(7):
            Tj = np.eye(4); Tj[:3, :3] = rpy(arm["joint_rpy"][i]); Tj[:3, 3] = arm["joint_xyz"][i]
            T = T @ Tj
            self.frames.append(T.copy()); self.parent.append(i - 1); self.axis.append((0, 0, 1))
            self.lim.append((arm["joint_lower"][i], arm["joint_upper"][i])); self.damp.append(arm["joint_damping"]); self.fixed.append(False)
            self.dyn.append((arm["mass"][i], arm["com"][i], arm["inertia"][i]))
        self.dyn[6] = (0.3, np.array([0, 0, 0.02]), np.array([0.001] * 3))
        z7 = 1.261

        def world(pose):                     # an sdf link pose (model frame, q = 0) -> frame relative to link_7 -> world of this fake
            Tl = np.eye(4); Tl[:3, :3] = rpy(pose[3:]); Tl[:3, 3] = np.array(pose[:3]) - np.array([0, 0, z7])
            return self.frames[6] @ Tl
        sdf = [((0, 0, 1.305, 0, 0, 0), (0, 0, 0), 1.2, (1.0,) * 3, 6, (0, 0, 1), False, 0.5),
               ((0, 0.024, 1.35, 0, -0.05, 0), (0, 0, 0.04), 0.2, (0.1,) * 3, 7, (0, 1, 0), False, 0.5),
               ((-0.005, 0.024, 1.43, 0, -0.3, 0), (-0.003, 0, 0.04), 0.2, (0.1,) * 3, 8, (0, 0, 0), True, 0.8),
               ((-0.02, 0.024, 1.49, 0, 0.2, 0), (-0.005, 0, 0.026), 0.2, (0.1,) * 3, 9, (0, 1, 0), False, 0.8),
               ((0, -0.024, 1.35, 0, 0.05, 0), (0, 0, 0.04), 0.2, (0.1,) * 3, 7, (0, 1, 0), False, 0.5),
               ((0.005, -0.024, 1.43, 0, 0.3, 0), (0.003, 0, 0.04), 0.2, (0.1,) * 3, 11, (0, 0, 0), True, 0.8),
               ((0.02, -0.024, 1.49, 0, -0.2, 0), (0.005, 0, 0.026), 0.2, (0.1,) * 3, 12, (0, 1, 0), False, 0.8)]
        self.mu = [0.5] * 7
        for pose, ipos, mass, I, par, ax, fixed, mu in sdf:
            self.frames.append(world(pose)); self.parent.append(par); self.axis.append(ax); self.lim.append((-10.4, 10.01))
            self.damp.append(0.0); self.fixed.append(fixed); self.dyn.append((mass, np.array(ipos), np.array(I))); self.mu.append(mu)

    def getNumJoints(self, uid): return 14
    def getJointState(self, uid, j): return (0.0, 0.0)
    def resetJointState(self, uid, j, v): pass
    def getBasePositionAndOrientation(self, uid): return ((0.0, 0.0, 0.0), (0.0, 0.0, 0.0, 1.0))

    def getJointInfo(self, uid, j):
        info = [None] * 17
        info[2] = self.JOINT_FIXED if self.fixed[j] else 0
        info[6], info[8], info[9], info[13], info[16] = self.damp[j], self.lim[j][0], self.lim[j][1], self.axis[j], self.parent[j]
        return info

    @staticmethod
    def _quat(R):
        w = np.sqrt(max(0.0, 1 + R[0, 0] + R[1, 1] + R[2, 2])) / 2
        if w > 1e-8:
            return ((R[2, 1] - R[1, 2]) / (4 * w), (R[0, 2] - R[2, 0]) / (4 * w), (R[1, 0] - R[0, 1]) / (4 * w), w)
        x = np.sqrt(max(0.0, 1 + R[0, 0] - R[1, 1] - R[2, 2])) / 2                 # half turns of the iiwa chain: w = 0
        if x > 1e-8:
            return (x, (R[0, 1] + R[1, 0]) / (4 * x), (R[0, 2] + R[2, 0]) / (4 * x), (R[2, 1] - R[1, 2]) / (4 * x))
        y = np.sqrt(max(0.0, 1 - R[0, 0] + R[1, 1] - R[2, 2])) / 2
        if y > 1e-8:
            return ((R[0, 1] + R[1, 0]) / (4 * y), y, (R[1, 2] + R[2, 1]) / (4 * y), (R[0, 2] - R[2, 0]) / (4 * y))
        z = np.sqrt(max(0.0, 1 - R[0, 0] - R[1, 1] + R[2, 2])) / 2
        return ((R[0, 2] + R[2, 0]) / (4 * z), (R[1, 2] + R[2, 1]) / (4 * z), z, (R[1, 0] - R[0, 1]) / (4 * z))

    def getLinkState(self, uid, j, computeForwardKinematics=True):
        T = self.frames[j]
        return (None, None, None, None, tuple(T[:3, 3]), self._quat(T[:3, :3]))

    def getDynamicsInfo(self, uid, j):
        mass, c, I = self.dyn[j]
        return (mass, self.mu[j], tuple(I), tuple(c), (0.0, 0.0, 0.0, 1.0))


def test_tree_from_pybullet_on_a_fake_that_serves_the_recalled_sdf():
    base = kuka_model.tree_default()
    m = kuka_model.tree_from_pybullet(FakeBullet(), 0, base=base)
    got, want = kuka_model.tree_to_table(m), kuka_model.tree_to_table(base)
    # joint 0's origin: the fake has no base offset, the table adds kuka.py:63's base position in the kernel, not in the table
    assert np.abs(got - want).max() < 1e-9, np.nonzero(np.abs(got - want) > 1e-9)
