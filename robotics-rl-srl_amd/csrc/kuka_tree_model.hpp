// kuka_tree_model.hpp — the FULL Kuka model as data: the 12-DoF tree of pybullet_data/kuka_iiwa/kuka_with_gripper2.sdf (kuka.py:60)
//   DoF 0..6   arm joints J0..J6 (joint index 0..6)
//   DoF 7      gripper_to_arm, continuous about z (joint 7), child base_link
//   DoF 8, 9   base_left_finger_joint (8) -> left_base_tip_joint (10); the fixed joint 9 (left_finger -> left_finger_base) is merged
//   DoF 10, 11 base_right_finger_joint (11) -> right_base_tip_joint (13); fixed joint 12 merged
// with the POSITION_CONTROL motor the reference commands on every joint each step (kuka.py:167-187), 16 collision spheres on links
// 5..11 and one friction direction per contact.  `srlhip_kuka_tree_model` of include/srlhip.h has this layout (510 doubles, integer
// fields stored as doubles).  The arm part repeats kuka_core.hpp's table; the gripper part is RECALLED from the SDF file
// [UNVERIFIED-MEMORY] (pybullet_data is absent here: PARITY UNPINNED) — which is why it is a runtime table:
// tests/golden/make_kuka_pybullet_golden.py fills it from the real files.
#pragma once
#include <math.h>
#include <string.h>

#include "kuka_core.hpp"

namespace srl {
namespace kuka {

constexpr int TN = 12, TNS = 16;
struct TreeJoint {
    double parent, xyz[3], Rj[9] /* row-major: child = Rj * Rot(axis, q) in the parent link's frame */, axis[3], lower, upper /* lower > upper: none */,
        damping, mass, com[3], inertia[6] /* about the COM, link axes: xx xy xz yy yz zz */, kp, max_force, max_vel, joint_index;
};
struct TreeSphere { double link, c[3], r, mu; };
struct TreeModel {
    double nd;
    TreeJoint j[TN];
    double ee_link, ee_point[3], grip_link, grip_point[3], nsphere;
    TreeSphere s[TNS];
    double table_top_z, button_base_z, max_generic_rows, friction;
    // solver details of the dependency that are recalled, not read — the PyBullet pin decides them as data (include/srlhip.h)
    double solver_detail, contact_erp, limit_erp, linear_slop;
};
constexpr int kTreeModelDoubles = 510;
constexpr int kDetailAltSweep = 1, kDetailBodyOrder = 2, kDetailFriction2 = 4;       // SRLHIP_KUKA_DETAIL_*
static_assert(sizeof(TreeModel) == kTreeModelDoubles * sizeof(double), "srlhip_kuka_tree_model layout");

namespace tree_build {
struct SdfLink { double pose[6], ipos[3], mass, inertia[3]; };
// kuka_with_gripper2.sdf: link pose in the model frame at q = 0 (xyz, rpy), inertial offset, mass, diagonal inertia [UNVERIFIED-MEMORY]
constexpr SdfLink kLink7 = {{0, 0, 1.261, 0, 0, 0}, {0, 0, 0.02}, 0.3, {0.001, 0.001, 0.001}};
constexpr SdfLink kBaseLink = {{0, 0, 1.305, 0, 0, 0}, {0, 0, 0}, 1.2, {1.0, 1.0, 1.0}};
constexpr SdfLink kFinger[2] = {{{0, 0.024, 1.35, 0, -0.05, 0}, {0, 0, 0.04}, 0.2, {0.1, 0.1, 0.1}}, {{0, -0.024, 1.35, 0, 0.05, 0}, {0, 0, 0.04}, 0.2, {0.1, 0.1, 0.1}}};
constexpr SdfLink kFingerBase[2] = {{{-0.005, 0.024, 1.43, 0, -0.3, 0}, {-0.003, 0, 0.04}, 0.2, {0.1, 0.1, 0.1}},
                                    {{0.005, -0.024, 1.43, 0, 0.3, 0}, {0.003, 0, 0.04}, 0.2, {0.1, 0.1, 0.1}}};
constexpr SdfLink kTip[2] = {{{-0.02, 0.024, 1.49, 0, 0.2, 0}, {-0.005, 0, 0.026}, 0.2, {0.1, 0.1, 0.1}}, {{0.02, -0.024, 1.49, 0, -0.2, 0}, {0.005, 0, 0.026}, 0.2, {0.1, 0.1, 0.1}}};
constexpr double kGripperKp = 0.1, kFingerAForce = 2.0, kFingerBForce = 2.5, kFingerTipForce = 2.0;   // kuka.py:33-36, pybullet default positionGain
constexpr double kMuDefault = 0.25, kMuFinger = 0.4;      // 0.5 x 0.5 ; finger base / tip links 0.8 x 0.5

inline void rpy_rows(const double rpy[3], double R[9]) {          // row-major Rz(yaw) Ry(pitch) Rx(roll)
    const double cr = cos(rpy[0]), sr = sin(rpy[0]), cp = cos(rpy[1]), sp = sin(rpy[1]), cy = cos(rpy[2]), sy = sin(rpy[2]);
    R[0] = cy * cp; R[1] = cy * sp * sr - sy * cr; R[2] = cy * sp * cr + sy * sr;
    R[3] = sy * cp; R[4] = sy * sp * sr + cy * cr; R[5] = sy * sp * cr - cy * sr;
    R[6] = -sp;     R[7] = cp * sr;                R[8] = cp * cr;
}
inline void relative(const double parent[6], const double child[6], double xyz[3], double R[9]) {   // child pose in the parent's frame
    double Rp[9], Rc[9], d[3];
    rpy_rows(parent + 3, Rp); rpy_rows(child + 3, Rc);
    for (int k = 0; k < 3; k++) d[k] = child[k] - parent[k];
    for (int i = 0; i < 3; i++) xyz[i] = Rp[i] * d[0] + Rp[3 + i] * d[1] + Rp[6 + i] * d[2];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double s = 0; for (int k = 0; k < 3; k++) s += Rp[3 * k + i] * Rc[3 * k + j]; R[3 * i + j] = s; }
}
// inertial parameters of a body = one SDF link, optionally with a second one welded to it (fixed joint): exact composite
inline void set_body(TreeJoint &J, const SdfLink &a, const SdfLink *b) {
    if (!b) {
        J.mass = a.mass;
        for (int k = 0; k < 3; k++) J.com[k] = a.ipos[k];
        J.inertia[0] = a.inertia[0]; J.inertia[3] = a.inertia[1]; J.inertia[5] = a.inertia[2]; J.inertia[1] = J.inertia[2] = J.inertia[4] = 0.0;
        return;
    }
    double xyz[3], R[9], cb[3], I[3][3] = {{0}};
    relative(a.pose, b->pose, xyz, R);
    for (int i = 0; i < 3; i++) cb[i] = xyz[i] + R[3 * i] * b->ipos[0] + R[3 * i + 1] * b->ipos[1] + R[3 * i + 2] * b->ipos[2];
    const double mt = a.mass + b->mass;
    double c[3];
    for (int k = 0; k < 3; k++) c[k] = (a.mass * a.ipos[k] + b->mass * cb[k]) / mt;
    for (int i = 0; i < 3; i++) I[i][i] = a.inertia[i];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double s = 0; for (int k = 0; k < 3; k++) s += R[3 * i + k] * b->inertia[k] * R[3 * j + k]; I[i][j] += s; }
    const double *cs[2] = {a.ipos, cb}; const double ms[2] = {a.mass, b->mass};
    for (int s = 0; s < 2; s++) {
        double r[3]; for (int k = 0; k < 3; k++) r[k] = cs[s][k] - c[k];
        const double rr = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) I[i][j] += ms[s] * ((i == j ? rr : 0.0) - r[i] * r[j]);
    }
    J.mass = mt;
    for (int k = 0; k < 3; k++) J.com[k] = c[k];
    J.inertia[0] = I[0][0]; J.inertia[1] = I[0][1]; J.inertia[2] = I[0][2]; J.inertia[3] = I[1][1]; J.inertia[4] = I[1][2]; J.inertia[5] = I[2][2];
}
inline void add_sphere(TreeModel &m, int link, const double *fxyz, const double *fR, double x, double y, double z, double r, double mu) {
    TreeSphere &s = m.s[(int)m.nsphere]; m.nsphere += 1;
    const double c[3] = {x, y, z};
    s.link = link; s.r = r; s.mu = mu;
    for (int i = 0; i < 3; i++) s.c[i] = fxyz ? fxyz[i] + fR[3 * i] * c[0] + fR[3 * i + 1] * c[1] + fR[3 * i + 2] * c[2] : c[i];
}
}  // namespace tree_build

inline void default_tree_model(TreeModel &m) {
    using namespace tree_build;
    static const int jidx[TN] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 11, 13};
    memset(&m, 0, sizeof m);
    Model arm; default_model(arm);
    m.nd = 12;
    for (int i = 0; i < ND; i++) {
        TreeJoint &J = m.j[i];
        J.parent = i - 1;
        for (int k = 0; k < 3; k++) { J.xyz[k] = arm.joint_xyz[i][k]; J.com[k] = arm.com[i][k]; J.axis[k] = k == 2 ? 1.0 : 0.0; }
        rpy_rows(arm.joint_rpy[i], J.Rj);
        J.lower = arm.joint_lower[i]; J.upper = arm.joint_upper[i]; J.damping = arm.joint_damping; J.mass = arm.mass[i];
        J.inertia[0] = arm.inertia[i][0]; J.inertia[3] = arm.inertia[i][1]; J.inertia[5] = arm.inertia[i][2];
        J.kp = kArmKp; J.max_force = kArmMaxForce; J.max_vel = kArmMaxVel;
    }
    set_body(m.j[6], kLink7, nullptr);                         // link_7 without the lumped gripper
    m.j[7].parent = 6; relative(kLink7.pose, kBaseLink.pose, m.j[7].xyz, m.j[7].Rj);
    m.j[7].axis[2] = 1.0; set_body(m.j[7], kBaseLink, nullptr);
    m.j[7].kp = kGripperKp; m.j[7].max_force = kArmMaxForce;
    for (int side = 0; side < 2; side++) {
        const int f = 8 + 2 * side, t = f + 1;
        m.j[f].parent = 7; relative(kBaseLink.pose, kFinger[side].pose, m.j[f].xyz, m.j[f].Rj);
        m.j[t].parent = f; relative(kFinger[side].pose, kTip[side].pose, m.j[t].xyz, m.j[t].Rj);
        m.j[f].axis[1] = 1.0; m.j[t].axis[1] = 1.0;
        set_body(m.j[f], kFinger[side], &kFingerBase[side]);
        set_body(m.j[t], kTip[side], nullptr);
        m.j[f].kp = kGripperKp; m.j[t].kp = kGripperKp;
        m.j[f].max_force = side == 0 ? kFingerAForce : kFingerBForce; m.j[t].max_force = kFingerTipForce;
    }
    for (int i = 7; i < TN; i++) { m.j[i].lower = 1.0; m.j[i].upper = -1.0; m.j[i].damping = 0.0; m.j[i].max_vel = 1e30; }
    for (int i = 0; i < TN; i++) m.j[i].joint_index = jidx[i];
    m.ee_link = 6; m.grip_link = 8;
    for (int k = 0; k < 3; k++) { m.ee_point[k] = kLink7.ipos[k]; m.grip_point[k] = kFinger[0].ipos[k]; }
    m.nsphere = 0;
    add_sphere(m, 5, nullptr, nullptr, 0, 0, 0, 0.07, kMuDefault);
    add_sphere(m, 6, nullptr, nullptr, 0, 0, 0.02, 0.05, kMuDefault);
    add_sphere(m, 7, nullptr, nullptr, 0, 0, -0.025, 0.035, kMuDefault);
    add_sphere(m, 7, nullptr, nullptr, 0, 0, 0.025, 0.035, kMuDefault);
    for (int side = 0; side < 2; side++) {
        const int f = 8 + 2 * side, t = f + 1;
        double xyz[3], R[9];
        add_sphere(m, f, nullptr, nullptr, 0, 0, 0.02, 0.008, kMuDefault);
        add_sphere(m, f, nullptr, nullptr, 0, 0, 0.06, 0.008, kMuDefault);
        relative(kFinger[side].pose, kFingerBase[side].pose, xyz, R);
        add_sphere(m, f, xyz, R, 0, 0, 0.015, 0.012, kMuFinger);
        add_sphere(m, f, xyz, R, 0, 0, 0.045, 0.012, kMuFinger);
        add_sphere(m, t, nullptr, nullptr, 0, 0, 0.012, 0.010, kMuFinger);
        add_sphere(m, t, nullptr, nullptr, 0, 0, 0.032, 0.010, kMuFinger);
    }
    m.table_top_z = kTableTopZ; m.button_base_z = kButtonBaseZ; m.max_generic_rows = 6; m.friction = 1;
    m.solver_detail = 0; m.contact_erp = kErp; m.limit_erp = kErp; m.linear_slop = 0.0;
}

}  // namespace kuka
}  // namespace srl
