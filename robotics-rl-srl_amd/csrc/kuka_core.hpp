// kuka_core.hpp — per-env physics of the KukaButtonGymEnv stepper (gfx950).
//
// One lane integrates one env in float64:
//   Kuka.applyAction           environments/kuka_gym/kuka.py:118-187
//   p.calculateInverseKinematics -> one damped-least-squares step (7x7 SPD solve)
//   p.stepSimulation           -> Featherstone ABA + motor / limit / contact rows
//                                 + 150 projected Gauss-Seidel sweeps + semi-implicit Euler
//   p.getContactPoints / getLinkState -> analytic gripper-sphere contacts, gripper position
//
// MI355X-first formulation (not the textbook link-frame recursion the CPU oracle uses):
//   * ABA runs in WORLD coordinates about the world origin, so the articulated
//     inertia of a child is accumulated into its parent by plain addition — no
//     6x6 congruence transforms.  The backward sweep walks the kinematic chain
//     down from link 7, un-rotating the frame with the joint's cached sin/cos,
//     so only the joint axes z_i, origins p_i and the per-link ABA factors
//     (U_i, 1/d_i, u_i) are staged in LDS; the 21-entry articulated inertia,
//     the 7x7 M^-1 (28 unique), and every PGS accumulator live in VGPRs.
//   * M^-1 = seven unit-torque ABA back/forward sweeps evaluated together
//     (7-way ILP) on the staged factors; motor rows have unit Jacobians, so a
//     Gauss-Seidel row update is one scalar chain plus one 7-wide FMA axpy.
//   * LDS layout is [slot][lane]: a wavefront's 64 lanes hit 64 consecutive
//     8-byte words -> conflict-free ds_read_b64 / ds_write_b64.
// Everything is `__host__ __device__` so the identical source is also compiled
// for the host by the CPU-side parity harness (csrc/kuka_hostcheck.cpp, tests only).
#pragma once
#include <math.h>
#include <stdint.h>

#include "rng.hpp"

namespace srl {
namespace kuka {

// ---------------------------------------------------------------- model table
// In-tree constants: kuka.py:21-53,63-73,167-187; kuka_button_gym_env.py:17-35,219-236,347;
// urdf/simple_button.urdf.  Recalled (unverifiable here) constants of
// pybullet_data/kuka_iiwa/kuka_with_gripper2.sdf: SURVEY.md App. B.4 / DESIGN.md §Kuka model.
constexpr int ND = 7;
constexpr double kPi = 3.14159265358979323846;
constexpr double kDt = 1.0 / 240.0;
#ifndef SRL_EXPERIMENT_SOLVER_ITERS
constexpr int kSolverIters = 150;
#else   // timing experiments only (never shipped): see DESIGN.md §Kuka kernel, phase split
constexpr int kSolverIters = SRL_EXPERIMENT_SOLVER_ITERS;
#endif
constexpr double kGravityZ = -10.0;
constexpr double kBasePos[3] = {-0.1, 0.0, -0.15};
// joint frame in the parent link: translation along one parent axis, then a fixed
// signed-permutation rotation (URDF rpy of the iiwa chain), then Rz(q).
//   fix 0: identity            fix 1: rpy (pi/2,0,pi) == (-pi/2,pi,0): [-x, z, y]
//   fix 2: rpy (pi/2,0,0): [x, z, -y]
constexpr int kFix[ND] = {0, 1, 1, 2, 1, 2, 1};
constexpr int kTransAxis[ND] = {2, 2, 1, 2, 1, 2, 1};          // 1 = parent y, 2 = parent z
constexpr double kTransLen[ND] = {0.1575, 0.2025, 0.2045, 0.2155, 0.1845, 0.2155, 0.081};
constexpr double kJointLower[ND] = {-2.96705972839, -2.09439510239, -2.96705972839, -2.09439510239,
                                    -2.96705972839, -2.09439510239, -3.05432619099};
constexpr double kJointUpper[ND] = {2.96705972839, 2.09439510239, 2.96705972839, 2.09439510239,
                                    2.96705972839, 2.09439510239, 3.05432619099};
constexpr double kJointDamping = 0.5;
constexpr double kMass[ND] = {4.0, 4.0, 3.0, 2.7, 1.7, 1.8, 1.8};
constexpr double kCom[ND][3] = {{0, -0.03, 0.12}, {0.0003, 0.059, 0.042}, {0, 0.03, 0.13}, {0, 0.067, 0.034},
                                {0.0001, 0.021, 0.076}, {0, 0.0006, 0.0004}, {0, 0, 0.31 / 3.0}};
constexpr double kInertia[ND][3] = {{0.1, 0.09, 0.02}, {0.05, 0.018, 0.044}, {0.08, 0.075, 0.01}, {0.03, 0.01, 0.029},
                                    {0.02, 0.018, 0.005}, {0.005, 0.0036, 0.0047}, {0.0075, 0.0075, 0.003}};
constexpr double kJointPositions[14] = {0.006418, 0.113184, -0.011401, -1.289317, 0.005379, 1.737684, -0.006539,
                                        0.000048, -0.299912, 0.000000, -0.000043, 0.299960, 0.000000, -0.000200};
constexpr double kEeInit[3] = {0.537, 0.0, 0.5};
constexpr double kEeBox[2][2][3] = {{{0.35, -0.30, 0.0}, {0.65, 0.30, 0.5}}, {{0.50, -0.17, 0.0}, {0.65, 0.22, 0.5}}};
constexpr double kArmKp = 0.3, kArmMaxVel = 0.35, kArmMaxForce = 200.0;
constexpr double kEePoint[3] = {0, 0, 0.02};
constexpr double kGripperPoint[3] = {0, 0.024, 0.10};
constexpr int kNSphere = 6;
constexpr double kSphere[kNSphere][4] = {{0, 0.020, 0.255, 0.015}, {0, -0.020, 0.255, 0.015}, {0, 0.035, 0.200, 0.020},
                                         {0, -0.035, 0.200, 0.020}, {0, 0, 0.100, 0.060}, {0, 0, 0.0, 0.070}};
constexpr double kIkDamping = 1e-5;
// Kuka2Button calls calculateInverseKinematics with 7-entry null-space lists on a 12-DoF body and without jointDamping
// (kuka.py:147-148): pybullet ignores null-space lists whose length is not the DoF count and then falls back to its
// default damping [UNVERIFIED-MEMORY, DESIGN.md §Kuka2Button]
constexpr double kIkDampingDefault = 0.5;
// IK conditioning flag (srlhip.h SRLHIP_KUKA_IK_CROSS_DET, oracle KM_IK_CROSS_DET): det(J^T J + damping I) of an IK solve below
// this marks the episode as having crossed the neighbourhood of a kinematic singularity, where the controller's closed loop
// amplifies float64 rounding noise by a constant factor per step (no two implementations agree to 1e-4 beyond it)
constexpr double kIkCrossDet = 3e-9;
constexpr double kButton1Y2B = 0.125, kButton2Y2B = -0.125, kZTable = -0.2;   // kuka_2button_gym_env.py:56-70
constexpr int kMaxSteps2Button = 1500;
constexpr double kIkMaxAngle = 45.0 * kPi / 180.0;
constexpr double kTableTopZ = -0.195, kButtonBaseZ = -0.195, kButtonX = 0.5, kButtonY = 0.0;
constexpr double kGliderOriginZ = 0.005, kGliderLower = 0.0, kGliderUpper = 0.01, kCapMass = 0.1;
constexpr double kCapRadius = 0.09, kCapHeight = 0.03, kBaseRadius = 0.10, kBaseHeight = 0.03;
constexpr double kButtonDistanceHeight = 0.28, kButtonTarget = 0.1, kButtonKp = 0.1, kButtonMaxForce = 100000.0;
constexpr double kDefaultMotorImpulse = 1.0, kLimitMaxImpulse = 100.0, kErp = 0.2, kContactThreshold = 0.002;
constexpr double kLimitActivationVel = 10.0;   // limit rows exist when stop distance / dt <= this
constexpr int kMaxSteps = 1000, kNContactsBeforeTermination = 5, kNStepsOutside = 5000;
constexpr double kDeltaV = 0.03, kDeltaVContinuous = 0.0035, kDeltaTheta = 0.1;
constexpr double kNoiseStd = 0.01, kNoiseStdContinuous = 0.0001, kNoiseStdJoints = 0.002;
constexpr int kNInitActions = 5, kNSettleSteps = 500;

// ---------------------------------------------------------------- runtime model table
// The RECALLED part of the model above as data: `srlhip_kuka_model` of include/srlhip.h (138 doubles, same layout as the
// oracle's table in oracle/kuka_model.h).  The lane-group kernel (kuka_group.hpp) can integrate any table of this shape — a
// 7-joint serial arm whose joints turn about their local z — so that link frames / inertial parameters extracted from
// pybullet_data's kuka_with_gripper2.sdf can be dropped in the moment PyBullet is available
// (tests/golden/make_kuka_pybullet_golden.py); the lane-per-env kernel is specialised for the baked table.
struct Model {
    double joint_xyz[ND][3], joint_rpy[ND][3], joint_lower[ND], joint_upper[ND], joint_damping;
    double mass[ND], com[ND][3], inertia[ND][3], ee_point[3], gripper_point[3], sphere[kNSphere][4], table_top_z, button_base_z;
};
static_assert(sizeof(Model) == 138 * sizeof(double), "srlhip_kuka_model layout");
inline void default_model(Model &m) {
    const double hp = kPi / 2;
    const double rpy[3][3] = {{0, 0, 0}, {hp, 0, kPi}, {hp, 0, 0}};        // kFix 0 / 1 / 2 (1 is also (-pi/2, pi, 0))
    for (int i = 0; i < ND; i++) {
        for (int k = 0; k < 3; k++) {
            m.joint_xyz[i][k] = k == kTransAxis[i] ? kTransLen[i] : 0.0;
            m.joint_rpy[i][k] = (kFix[i] == 1 && i >= 4) ? (k == 0 ? -hp : k == 1 ? kPi : 0.0) : rpy[kFix[i]][k];   // same rotation, the sdf's spelling
            m.com[i][k] = kCom[i][k]; m.inertia[i][k] = kInertia[i][k];
        }
        m.joint_lower[i] = kJointLower[i]; m.joint_upper[i] = kJointUpper[i]; m.mass[i] = kMass[i];
    }
    m.joint_damping = kJointDamping;
    for (int k = 0; k < 3; k++) { m.ee_point[k] = kEePoint[k]; m.gripper_point[k] = kGripperPoint[k]; }
    for (int s = 0; s < kNSphere; s++) for (int k = 0; k < 4; k++) m.sphere[s][k] = kSphere[s][k];
    m.table_top_z = kTableTopZ; m.button_base_z = kButtonBaseZ;
}

// ---------------------------------------------------------------- LDS scratch
// [slot][lane] doubles; `st` = lanes per workgroup (64 on the device, 1 on the host).
constexpr int kMaxGenRows = 6;       // arm-limit rows + contact rows kept per step (first come, first kept)
constexpr int SC_Z = 0;              // z_i      [7][3]
constexpr int SC_P = 21;             // p_i      [7][3]
constexpr int SC_U = 42;             // U_i      [7][6]
constexpr int SC_DINV = 84;          // 1/d_i    [7]
constexpr int SC_UU = 91;            // u_i      [7]
constexpr int SC_STAGE = 98;         // end of the per-link staging
// generic rows [kMaxGenRows][21]: J7 WJ7 Jb WJb Dinv rhs lo hi applied
constexpr int kLdsRows = 2;
constexpr int ROW_J = 0, ROW_WJ = 7, ROW_JB = 14, ROW_WJB = 15, ROW_DINV = 16, ROW_RHS = 17, ROW_LO = 18,
              ROW_HI = 19, ROW_APP = 20, ROW_STRIDE = 21;
constexpr int SC_TOTAL = SC_STAGE + kLdsRows * ROW_STRIDE;   // 140 doubles = 1120 B of LDS per env (70 KiB / workgroup, 2 per CU)
constexpr int SC_ROWS_TOTAL = kMaxGenRows * ROW_STRIDE;      // 126 doubles of global scratch per env (rows 2.. used)

struct Scratch {
    double *b;       // LDS base of this lane, [slot][lane]: per-link ABA staging
    int st;
    double *g;       // HBM (L2-resident) base of this env, [slot][env]: generic constraint rows.  They are written once
    int64_t gst;     // per physics step; the first two rows of a lane are then held in VGPRs for all 150 sweeps.
    double *objs;    // KukaRandButton: [30][env] (x, y, present) of the distractor objects, stride gst; may be null
    SRL_HD double &at(int i) const { return b[(int64_t)i * st]; }
    // rows 0 and 1 (the common contact case) are staged in LDS, rows 2.. in the global scratch
    SRL_HD double &row(int i) const { return i < kLdsRows * ROW_STRIDE ? b[(int64_t)(SC_STAGE + i) * st] : g[(int64_t)i * gst]; }
};

#if defined(__HIP_DEVICE_COMPILE__)
#define SRL_ANY(pred) (__any(pred))
#else
#define SRL_ANY(pred) (pred)
#endif

// ---------------------------------------------------------------- env state
struct Env {
    double q[ND], qd[ND];      // arm joint positions / velocities
    double sq[ND], cq[ND];     // sin/cos of q (cached: one sincos per joint per physics step)
    double ee[3];              // Kuka.end_effector_pos (IK target accumulator)
    double bq, bqd;            // button glider
    double bx, by, bz;         // button base position (bz moves only in the MovingButton env)
    double bspeed;             // MovingButton: signed y increment per step
    double bpos[3];            // button_pos (target point: cap + 0.28)
    double grip[3];            // getArmPos() after the last physics step
    int32_t motor_on;          // button motor: 0 pybullet default velocity motor, 1 position target (step2)
    int32_t contact_button, contact_table;
    int32_t counter, n_contacts, n_outside, terminated;
    // Kuka2ButtonGymEnv only (touched by the NB == 2 instantiations): second button, per-body contact flags
    // (getContactPoints(button_uid[k], kuka) has no link filter there), goal bookkeeping
    double b2q, b2qd, b2x, b2y;
    int32_t contact_body1, contact_body2, goal_id, n_contacts2;
    // bit 0: sticky per episode — an IK solve of this episode had det(J^T J + damping I) < kIkCrossDet; bits 1..: env-steps taken with
    // the bit set since the handle was created (full model only; the lumped kernels leave it 0)
    int32_t ikx;
};

struct Cfg {
    int32_t random_target, force_down, shape_reward, action_repeat, is_discrete, action_joints, obs_mode, auto_reset;
    int32_t moving, max_steps;  // KukaMovingButtonGymEnv: moving = 1, max_steps = 1500
    int32_t two;                // Kuka2ButtonGymEnv: large workspace, default IK damping, max_steps = 1500
    int32_t rand_objects;       // KukaRandButtonGymEnv: reset() draws ten distractor positions (scenery only)
    int32_t info_bits;          // srlhip_config.info_bits: 1 = bit 1 of every done_out byte carries the step's IK conditioning flag (full model)
    double max_distance;
};

SRL_HD void cross3(const double a[3], const double b[3], double o[3]) {
    o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}
SRL_HD double dot3(const double a[3], const double b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

// frame columns x = R[0..2], y = R[3..5], z = R[6..8]
template <int I>
SRL_HD void fk_forward(double R[9], double p[3], double s, double c) {
    const double *col = kTransAxis[I] == 2 ? R + 6 : R + 3;
#pragma unroll
    for (int k = 0; k < 3; k++) p[k] += kTransLen[I] * col[k];
    double c0[3], c1[3], c2[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        if (kFix[I] == 0) { c0[k] = R[k]; c1[k] = R[3 + k]; c2[k] = R[6 + k]; }
        else if (kFix[I] == 1) { c0[k] = -R[k]; c1[k] = R[6 + k]; c2[k] = R[3 + k]; }
        else { c0[k] = R[k]; c1[k] = R[6 + k]; c2[k] = -R[3 + k]; }
    }
#pragma unroll
    for (int k = 0; k < 3; k++) { R[k] = c * c0[k] + s * c1[k]; R[3 + k] = c * c1[k] - s * c0[k]; R[6 + k] = c2[k]; }
}
// inverse of fk_forward<I>: frame of link I -> frame of link I-1
template <int I>
SRL_HD void fk_backward(double R[9], double p[3], double s, double c) {
    double c0[3], c1[3], c2[3];
#pragma unroll
    for (int k = 0; k < 3; k++) { c0[k] = c * R[k] - s * R[3 + k]; c1[k] = s * R[k] + c * R[3 + k]; c2[k] = R[6 + k]; }
#pragma unroll
    for (int k = 0; k < 3; k++) {
        if (kFix[I] == 0) { R[k] = c0[k]; R[3 + k] = c1[k]; R[6 + k] = c2[k]; }
        else if (kFix[I] == 1) { R[k] = -c0[k]; R[6 + k] = c1[k]; R[3 + k] = c2[k]; }
        else { R[k] = c0[k]; R[6 + k] = c1[k]; R[3 + k] = -c2[k]; }
    }
    const double *col = kTransAxis[I] == 2 ? R + 6 : R + 3;
#pragma unroll
    for (int k = 0; k < 3; k++) p[k] -= kTransLen[I] * col[k];
}

// Kinematic sweep: stages z_i, p_i in LDS, leaves link 7's frame in (R, p) and the spatial
// velocity of link 7 about the world origin in (w, vo).
template <int I>
SRL_HD void fk_all_step(const Env &e, const Scratch &sc, double R[9], double p[3], double w[3], double vo[3]) {
    fk_forward<I>(R, p, e.sq[I], e.cq[I]);
    double s2[3];
    cross3(p, R + 6, s2);
#pragma unroll
    for (int k = 0; k < 3; k++) {
        sc.at(SC_Z + 3 * I + k) = R[6 + k];
        sc.at(SC_P + 3 * I + k) = p[k];
        w[k] += R[6 + k] * e.qd[I];
        vo[k] += s2[k] * e.qd[I];
    }
}
SRL_HD void fk_all(const Env &e, const Scratch &sc, double R[9], double p[3], double w[3], double vo[3]) {
    R[0] = 1; R[1] = 0; R[2] = 0; R[3] = 0; R[4] = 1; R[5] = 0; R[6] = 0; R[7] = 0; R[8] = 1;
#pragma unroll
    for (int k = 0; k < 3; k++) { p[k] = kBasePos[k]; w[k] = 0; vo[k] = 0; }
    fk_all_step<0>(e, sc, R, p, w, vo); fk_all_step<1>(e, sc, R, p, w, vo); fk_all_step<2>(e, sc, R, p, w, vo);
    fk_all_step<3>(e, sc, R, p, w, vo); fk_all_step<4>(e, sc, R, p, w, vo); fk_all_step<5>(e, sc, R, p, w, vo);
    fk_all_step<6>(e, sc, R, p, w, vo);
}
// frame of link 7 only (gripper position after integration)
SRL_HD void fk_tip(const Env &e, double R[9], double p[3]) {
    R[0] = 1; R[1] = 0; R[2] = 0; R[3] = 0; R[4] = 1; R[5] = 0; R[6] = 0; R[7] = 0; R[8] = 1;
#pragma unroll
    for (int k = 0; k < 3; k++) p[k] = kBasePos[k];
    fk_forward<0>(R, p, e.sq[0], e.cq[0]); fk_forward<1>(R, p, e.sq[1], e.cq[1]); fk_forward<2>(R, p, e.sq[2], e.cq[2]);
    fk_forward<3>(R, p, e.sq[3], e.cq[3]); fk_forward<4>(R, p, e.sq[4], e.cq[4]); fk_forward<5>(R, p, e.sq[5], e.cq[5]);
    fk_forward<6>(R, p, e.sq[6], e.cq[6]);
}
SRL_HD void tip_point(const double R[9], const double p[3], const double l[3], double o[3]) {
#pragma unroll
    for (int k = 0; k < 3; k++) o[k] = p[k] + R[k] * l[0] + R[3 + k] * l[1] + R[6 + k] * l[2];
}
SRL_HD void update_trig_and_gripper(Env &e) {
#pragma unroll
    for (int i = 0; i < ND; i++) sincos(e.q[i], &e.sq[i], &e.cq[i]);
    double R[9], p[3];
    fk_tip(e, R, p);
    tip_point(R, p, kGripperPoint, e.grip);
}

// ---------------------------------------------------------------- inverse kinematics
// One damped-least-squares step towards (target, orientation quat(euler(0,-pi,0))): kuka.py:144-156.
SRL_HD void ik_step(const Env &e, const Scratch &sc, const double R[9], const double p[3], double damping, double qdes[ND]) {
    double ee[3], J[6][ND], dS[6];
    tip_point(R, p, kEePoint, ee);
#pragma unroll
    for (int j = 0; j < ND; j++) {
        double z[3], d[3], c[3];
#pragma unroll
        for (int k = 0; k < 3; k++) { z[k] = sc.at(SC_Z + 3 * j + k); d[k] = ee[k] - sc.at(SC_P + 3 * j + k); }
        cross3(z, d, c);
#pragma unroll
        for (int k = 0; k < 3; k++) { J[k][j] = c[k]; J[3 + k][j] = z[k]; }
    }
#pragma unroll
    for (int k = 0; k < 3; k++) dS[k] = e.ee[k] - ee[k];
    // current orientation as a quaternion (x, y, z, w); rows of the rotation matrix: m[r][c] = R[3*c + r]
    double qx, qy, qz, qw;
    {
        const double m00 = R[0], m01 = R[3], m02 = R[6], m10 = R[1], m11 = R[4], m12 = R[7], m20 = R[2], m21 = R[5], m22 = R[8];
        const double tr = m00 + m11 + m22;
        if (tr > 0) { double s = sqrt(tr + 1.0) * 2; qw = 0.25 * s; qx = (m21 - m12) / s; qy = (m02 - m20) / s; qz = (m10 - m01) / s; }
        else if (m00 > m11 && m00 > m22) { double s = sqrt(1.0 + m00 - m11 - m22) * 2; qw = (m21 - m12) / s; qx = 0.25 * s; qy = (m01 + m10) / s; qz = (m02 + m20) / s; }
        else if (m11 > m22) { double s = sqrt(1.0 + m11 - m00 - m22) * 2; qw = (m02 - m20) / s; qx = (m01 + m10) / s; qy = 0.25 * s; qz = (m12 + m21) / s; }
        else { double s = sqrt(1.0 + m22 - m00 - m11) * 2; qw = (m10 - m01) / s; qx = (m02 + m20) / s; qy = (m12 + m21) / s; qz = 0.25 * s; }
    }
    {
        // deltaQ = target * current^-1 with target = (0, sin(-pi/2), 0, cos(-pi/2))
        const double tx = 0.0, ty = -1.0, tz = 0.0, tw = 6.123233995736766e-17;
        const double ix = -qx, iy = -qy, iz = -qz, iw = qw;
        double dw = tw * iw - tx * ix - ty * iy - tz * iz;
        double dx = tw * ix + tx * iw + ty * iz - tz * iy;
        double dy = tw * iy - tx * iz + ty * iw + tz * ix;
        double dz = tw * iz + tx * iy - ty * ix + tz * iw;
        // btQuaternion::getAngle()/getAxis() in the well-conditioned form 2*atan2(|xyz|, w), xyz/|xyz|
        const double sv = sqrt(dx * dx + dy * dy + dz * dz);
        double angle = 2.0 * atan2(sv, dw), ax, ay, az;
        if (sv * sv < 10.0 * 2.2204460492503131e-16) { ax = 1; ay = 0; az = 0; }
        else { ax = dx / sv; ay = dy / sv; az = dz / sv; }
        if (angle > kPi) angle -= 2 * kPi;
        dS[3] = angle * ax; dS[4] = angle * ay; dS[5] = angle * az;
    }
    // (J^T J + damping I) dtheta = J^T dS : symmetric positive definite -> LDL^T
    double A[ND][ND], b[ND];
#pragma unroll
    for (int i = 0; i < ND; i++) {
#pragma unroll
        for (int j = 0; j <= i; j++) {
            double s = 0;
#pragma unroll
            for (int k = 0; k < 6; k++) s += J[k][i] * J[k][j];
            A[i][j] = s;
        }
        A[i][i] += damping;
        double s = 0;
#pragma unroll
        for (int k = 0; k < 6; k++) s += J[k][i] * dS[k];
        b[i] = s;
    }
    double Dinv[ND];
#pragma unroll
    for (int j = 0; j < ND; j++) {          // A[i][j] (i>j) becomes L[i][j]; A[j][j] the pivot D_j
        double dj = A[j][j];
#pragma unroll
        for (int k = 0; k < j; k++) dj -= A[j][k] * A[j][k] * A[k][k];
        A[j][j] = dj;
        Dinv[j] = 1.0 / dj;
#pragma unroll
        for (int i = j + 1; i < ND; i++) {
            double s = A[i][j];
#pragma unroll
            for (int k = 0; k < j; k++) s -= A[i][k] * A[j][k] * A[k][k];
            A[i][j] = s * Dinv[j];
        }
    }
#pragma unroll
    for (int i = 0; i < ND; i++) {
#pragma unroll
        for (int k = 0; k < i; k++) b[i] -= A[i][k] * b[k];
    }
#pragma unroll
    for (int i = 0; i < ND; i++) b[i] *= Dinv[i];
#pragma unroll
    for (int i = ND - 1; i >= 0; i--) {
#pragma unroll
        for (int k = i + 1; k < ND; k++) b[i] -= A[k][i] * b[k];
    }
    double maxabs = 0;
#pragma unroll
    for (int i = 0; i < ND; i++) maxabs = fmax(maxabs, fabs(b[i]));
    const double scale = maxabs > kIkMaxAngle ? kIkMaxAngle / maxabs : 1.0;
#pragma unroll
    for (int i = 0; i < ND; i++) qdes[i] = e.q[i] + (maxabs > kIkMaxAngle ? b[i] * scale : b[i]);
}

// ---------------------------------------------------------------- ABA (world frame, about the origin)
struct ArtInertia {          // [[A, H], [H^T, M]], A and M symmetric
    double A[6];             // xx xy xz yy yz zz
    double H[9];             // row-major 3x3
    double M[6];
};
SRL_HD void sym_mul(const double S[6], const double v[3], double o[3]) {
    o[0] = S[0] * v[0] + S[1] * v[1] + S[2] * v[2];
    o[1] = S[1] * v[0] + S[3] * v[1] + S[4] * v[2];
    o[2] = S[2] * v[0] + S[4] * v[1] + S[5] * v[2];
}
SRL_HD void sym_rank1_sub(double S[6], const double a[3], double k) {   // S -= k a a^T
    S[0] -= k * a[0] * a[0]; S[1] -= k * a[0] * a[1]; S[2] -= k * a[0] * a[2];
    S[3] -= k * a[1] * a[1]; S[4] -= k * a[1] * a[2]; S[5] -= k * a[2] * a[2];
}

template <int I>
SRL_HD void aba_backward_step(const Env &e, const Scratch &sc, double R[9], double p[3], double w[3], double vo[3],
                              ArtInertia &IA, double pAw[3], double pAv[3]) {
    // -- rigid-body inertia of link I in world coordinates about the world origin
    const double m = kMass[I];
    double cw[3], Io[6], h[3];
#pragma unroll
    for (int k = 0; k < 3; k++) cw[k] = p[k] + R[k] * kCom[I][0] + R[3 + k] * kCom[I][1] + R[6 + k] * kCom[I][2];
    const double cc = dot3(cw, cw);
    {
        const double ix = kInertia[I][0], iy = kInertia[I][1], iz = kInertia[I][2];
        Io[0] = ix * R[0] * R[0] + iy * R[3] * R[3] + iz * R[6] * R[6] + m * (cc - cw[0] * cw[0]);
        Io[1] = ix * R[0] * R[1] + iy * R[3] * R[4] + iz * R[6] * R[7] - m * cw[0] * cw[1];
        Io[2] = ix * R[0] * R[2] + iy * R[3] * R[5] + iz * R[6] * R[8] - m * cw[0] * cw[2];
        Io[3] = ix * R[1] * R[1] + iy * R[4] * R[4] + iz * R[7] * R[7] + m * (cc - cw[1] * cw[1]);
        Io[4] = ix * R[1] * R[2] + iy * R[4] * R[5] + iz * R[7] * R[8] - m * cw[1] * cw[2];
        Io[5] = ix * R[2] * R[2] + iy * R[5] * R[5] + iz * R[8] * R[8] + m * (cc - cw[2] * cw[2]);
    }
#pragma unroll
    for (int k = 0; k < 3; k++) h[k] = m * cw[k];
#pragma unroll
    for (int k = 0; k < 6; k++) IA.A[k] += Io[k];
    IA.H[1] -= h[2]; IA.H[2] += h[1]; IA.H[3] += h[2]; IA.H[5] -= h[0]; IA.H[6] -= h[1]; IA.H[7] += h[0];
    IA.M[0] += m; IA.M[3] += m; IA.M[5] += m;
    // -- velocity-product force of the link: v x* (I v)
    {
        double n[3], f[3], t0[3], t1[3];
        sym_mul(Io, w, n);
        cross3(h, vo, t0); cross3(h, w, t1);
#pragma unroll
        for (int k = 0; k < 3; k++) { n[k] += t0[k]; f[k] = m * vo[k] - t1[k]; }
        cross3(w, n, t0); cross3(vo, f, t1);
#pragma unroll
        for (int k = 0; k < 3; k++) pAw[k] += t0[k] + t1[k];
        cross3(w, f, t0);
#pragma unroll
        for (int k = 0; k < 3; k++) pAv[k] += t0[k];
    }
    // -- joint axis S = [z ; p x z], U = IA S, d = S.U, u = tau - S.pA
    const double *z = R + 6;
    double s2[3], Uw[3], Uv[3], t[3];
    cross3(p, z, s2);
    sym_mul(IA.A, z, Uw);
#pragma unroll
    for (int k = 0; k < 3; k++) Uw[k] += IA.H[3 * k] * s2[0] + IA.H[3 * k + 1] * s2[1] + IA.H[3 * k + 2] * s2[2];
    sym_mul(IA.M, s2, Uv);
#pragma unroll
    for (int k = 0; k < 3; k++) Uv[k] += IA.H[k] * z[0] + IA.H[3 + k] * z[1] + IA.H[6 + k] * z[2];
    const double d = dot3(z, Uw) + dot3(s2, Uv);
    const double dinv = 1.0 / d;
    const double tau = -kJointDamping * e.qd[I];
    const double u = tau - (dot3(z, pAw) + dot3(s2, pAv));
#pragma unroll
    for (int k = 0; k < 3; k++) { sc.at(SC_U + 6 * I + k) = Uw[k]; sc.at(SC_U + 6 * I + 3 + k) = Uv[k]; }
    sc.at(SC_DINV + I) = dinv;
    sc.at(SC_UU + I) = u;
    if (I > 0) {
        // bias acceleration c = v x (S qd)
        double cwv[3], cvv[3], t2[3];
        cross3(w, z, cwv); cross3(w, s2, cvv); cross3(vo, z, t2);
#pragma unroll
        for (int k = 0; k < 3; k++) { cwv[k] *= e.qd[I]; cvv[k] = (cvv[k] + t2[k]) * e.qd[I]; }
        // Ia = IA - U U^T / d
        sym_rank1_sub(IA.A, Uw, dinv);
        sym_rank1_sub(IA.M, Uv, dinv);
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
            for (int b = 0; b < 3; b++) IA.H[3 * a + b] -= dinv * Uw[a] * Uv[b];
        // pa = pA + Ia c + U u / d
        const double ud = u * dinv;
        sym_mul(IA.A, cwv, t);
#pragma unroll
        for (int k = 0; k < 3; k++)
            pAw[k] += t[k] + IA.H[3 * k] * cvv[0] + IA.H[3 * k + 1] * cvv[1] + IA.H[3 * k + 2] * cvv[2] + Uw[k] * ud;
        sym_mul(IA.M, cvv, t);
#pragma unroll
        for (int k = 0; k < 3; k++)
            pAv[k] += t[k] + IA.H[k] * cwv[0] + IA.H[3 + k] * cwv[1] + IA.H[6 + k] * cwv[2] + Uv[k] * ud;
        // walk the chain down: velocity and frame of the parent link
#pragma unroll
        for (int k = 0; k < 3; k++) { w[k] -= z[k] * e.qd[I]; vo[k] -= s2[k] * e.qd[I]; }
        fk_backward<I>(R, p, e.sq[I], e.cq[I]);
    }
}

// qdd (pass 3) and W = M^-1 (seven unit-torque sweeps evaluated side by side)
SRL_HD void aba_forward_and_minv(const Env &e, const Scratch &sc, double qdd[ND], double W[ND][ND]) {
    {
        double aw[3] = {0, 0, 0}, av[3] = {0, 0, -kGravityZ}, w[3] = {0, 0, 0}, vo[3] = {0, 0, 0};
#pragma unroll
        for (int i = 0; i < ND; i++) {
            double z[3], pp[3], s2[3], t0[3], t1[3], t2[3];
#pragma unroll
            for (int k = 0; k < 3; k++) { z[k] = sc.at(SC_Z + 3 * i + k); pp[k] = sc.at(SC_P + 3 * i + k); }
            cross3(pp, z, s2);
#pragma unroll
            for (int k = 0; k < 3; k++) { w[k] += z[k] * e.qd[i]; vo[k] += s2[k] * e.qd[i]; }
            cross3(w, z, t0); cross3(w, s2, t1); cross3(vo, z, t2);
            double Ua = 0;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                aw[k] += t0[k] * e.qd[i];
                av[k] += (t1[k] + t2[k]) * e.qd[i];
            }
#pragma unroll
            for (int k = 0; k < 3; k++) Ua += sc.at(SC_U + 6 * i + k) * aw[k] + sc.at(SC_U + 6 * i + 3 + k) * av[k];
            qdd[i] = (sc.at(SC_UU + i) - Ua) * sc.at(SC_DINV + i);
#pragma unroll
            for (int k = 0; k < 3; k++) { aw[k] += z[k] * qdd[i]; av[k] += s2[k] * qdd[i]; }
        }
    }
    // backward sweeps: column j starts at link j with u_j = 1
    double pw[ND][3], pv[ND][3], uu[ND][ND];
#pragma unroll
    for (int j = 0; j < ND; j++)
#pragma unroll
        for (int k = 0; k < 3; k++) { pw[j][k] = 0; pv[j][k] = 0; }
#pragma unroll
    for (int i = ND - 1; i >= 0; i--) {
        double z[3], pp[3], s2[3], Uw[3], Uv[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            z[k] = sc.at(SC_Z + 3 * i + k); pp[k] = sc.at(SC_P + 3 * i + k);
            Uw[k] = sc.at(SC_U + 6 * i + k); Uv[k] = sc.at(SC_U + 6 * i + 3 + k);
        }
        cross3(pp, z, s2);
        const double dinv = sc.at(SC_DINV + i);
#pragma unroll
        for (int j = 0; j < ND; j++) {
            if (j < i) { uu[i][j] = 0; continue; }          // torque on a descendant of link i only
            double u = (j == i ? 1.0 : 0.0) - (dot3(z, pw[j]) + dot3(s2, pv[j]));
            uu[i][j] = u;
            const double ud = u * dinv;
#pragma unroll
            for (int k = 0; k < 3; k++) { pw[j][k] += Uw[k] * ud; pv[j][k] += Uv[k] * ud; }
        }
    }
#pragma unroll
    for (int j = 0; j < ND; j++)
#pragma unroll
        for (int k = 0; k < 3; k++) { pw[j][k] = 0; pv[j][k] = 0; }     // now the link accelerations
#pragma unroll
    for (int i = 0; i < ND; i++) {
        double z[3], pp[3], s2[3], Uw[3], Uv[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            z[k] = sc.at(SC_Z + 3 * i + k); pp[k] = sc.at(SC_P + 3 * i + k);
            Uw[k] = sc.at(SC_U + 6 * i + k); Uv[k] = sc.at(SC_U + 6 * i + 3 + k);
        }
        cross3(pp, z, s2);
        const double dinv = sc.at(SC_DINV + i);
#pragma unroll
        for (int j = 0; j < ND; j++) {
            const double a = (uu[i][j] - (dot3(Uw, pw[j]) + dot3(Uv, pv[j]))) * dinv;
            W[i][j] = a;
#pragma unroll
            for (int k = 0; k < 3; k++) { pw[j][k] += z[k] * a; pv[j][k] += s2[k] * a; }
        }
    }
}

// ---------------------------------------------------------------- contacts
SRL_HD double sphere_cylinder(const double c[3], double rad, double bx, double by, double Rc, double z0, double z1,
                              double n[3]) {
    // The distance that is compared with kContactThreshold decides reward / done flags: evaluated UNFUSED, operation for operation as
    // oracle/kuka_oracle.c::sphere_cylinder does (-ffp-contract=off there), whatever the translation unit's contraction setting is.
    // (Its input, the sphere centre, still comes out of differently associated kinematics: flags are bit-exact as long as no
    // distance comes within ~1e-10 of the threshold — the oracle's margin probe reports how close the tested runs get.)
#pragma clang fp contract(off)
    const double dx = c[0] - bx, dy = c[1] - by, rho = sqrt(dx * dx + dy * dy);
    const double er = rho - Rc, ez_top = c[2] - z1, ez_bot = z0 - c[2];
    const double ez = ez_top > ez_bot ? ez_top : ez_bot;
    const double rx = rho > 1e-12 ? dx / rho : 1.0, ry = rho > 1e-12 ? dy / rho : 0.0, sz = ez_top > ez_bot ? 1.0 : -1.0;
    if (er <= 0 && ez <= 0) {
        if (er > ez) { n[0] = rx; n[1] = ry; n[2] = 0; return er - rad; }
        n[0] = 0; n[1] = 0; n[2] = sz; return ez - rad;
    }
    if (er <= 0) { n[0] = 0; n[1] = 0; n[2] = sz; return ez - rad; }
    if (ez <= 0) { n[0] = rx; n[1] = ry; n[2] = 0; return er - rad; }
    const double dist = sqrt(er * er + ez * ez);
    n[0] = rx * er / dist; n[1] = ry * er / dist; n[2] = sz * ez / dist;
    return dist - rad;
}

// generic (LDS) constraint row: arm Jacobian J, button Jacobian Jb
SRL_HD void add_generic_row(const Scratch &sc, int &ngen, const double J[ND], double Jb, const double W[ND][ND],
                            double desired, double pos_err, const Env &e, double bqd, double lo, double hi) {
    if (ngen >= kMaxGenRows) return;
    const int base = ngen * ROW_STRIDE;
    ngen++;
    double D = 0, rel = 0;
#pragma unroll
    for (int i = 0; i < ND; i++) {
        double s = 0;
#pragma unroll
        for (int j = 0; j < ND; j++) s += W[i][j] * J[j];
        sc.row(base + ROW_WJ + i) = s;
        sc.row(base + ROW_J + i) = J[i];
        D += J[i] * s;
        rel += J[i] * e.qd[i];
    }
    const double wjb = Jb * (1.0 / kCapMass);
    D += Jb * wjb;
    rel += Jb * bqd;              // velocity of the glider this row acts on
    const double dinv = 1.0 / D;
    sc.row(base + ROW_JB) = Jb; sc.row(base + ROW_WJB) = wjb; sc.row(base + ROW_DINV) = dinv;
    sc.row(base + ROW_RHS) = (desired - rel) * dinv + pos_err * dinv;
    sc.row(base + ROW_LO) = lo; sc.row(base + ROW_HI) = hi; sc.row(base + ROW_APP) = 0.0;
}

// One Gauss-Seidel update of LDS row k in impulse space: the arm velocity change so far is
// W lam + g with g = sum_l WJ_l mu_l, so J_k . dv = WJ_k . lam + J_k . g (+ the button part).
template <int NB>
SRL_HD void pgs_generic_row(const Scratch &sc, int k, const double lam[ND], double g[ND], double &dvb1, double &dvb2, uint32_t bsel) {
    const int base = k * ROW_STRIDE;
    const bool second = NB == 2 && ((bsel >> k) & 1u);      // the row's scalar Jacobian acts on button 2's glider
    double dvb = second ? dvb2 : dvb1;
    double jdv = sc.row(base + ROW_JB) * dvb;
#pragma unroll
    for (int i = 0; i < ND; i++) jdv += sc.row(base + ROW_WJ + i) * lam[i] + sc.row(base + ROW_J + i) * g[i];
    const double mu = sc.row(base + ROW_APP);
    const double sum = mu + (sc.row(base + ROW_RHS) - jdv * sc.row(base + ROW_DINV));
    const double cl = fmin(fmax(sum, sc.row(base + ROW_LO)), sc.row(base + ROW_HI));
    const double delta = cl - mu;
    sc.row(base + ROW_APP) = cl;
#pragma unroll
    for (int i = 0; i < ND; i++) g[i] += delta * sc.row(base + ROW_WJ + i);
    dvb += delta * sc.row(base + ROW_WJB);
    if (second) dvb2 = dvb; else dvb1 = dvb;
}

// ---------------------------------------------------------------- one physics step
// Kuka.applyAction (kuka.py:118-187) followed by p.stepSimulation().
template <int NB>
SRL_HD void physics_step(Env &e, const Cfg &cfg, const Scratch &sc, const double motor[3], bool joint_mode,
                         const double joint_targets[ND]) {
    const double dt = kDt;
    double R[9], p[3], w[3], vo[3], qdes[ND];
    fk_all(e, sc, R, p, w, vo);
    if (!joint_mode) {
        const int b = (cfg.random_target || cfg.two) ? 0 : 1;      // Kuka(small_constraints=False)
#pragma unroll
        for (int k = 0; k < 3; k++) {
            double v = e.ee[k] + motor[k];
            v = v < kEeBox[b][0][k] ? kEeBox[b][0][k] : v;
            v = v > kEeBox[b][1][k] ? kEeBox[b][1][k] : v;
            e.ee[k] = v;
        }
        ik_step(e, sc, R, p, cfg.two ? kIkDampingDefault : kIkDamping, qdes);
    } else {
#pragma unroll
        for (int i = 0; i < ND; i++) qdes[i] = joint_targets[i];
    }
    // -- collision detection at the current poses: gripper spheres vs cap / base / table
    double cc[kNSphere][3];
#pragma unroll
    for (int s = 0; s < kNSphere; s++) tip_point(R, p, kSphere[s], cc[s]);
    const double cap_z0 = e.bz + kGliderOriginZ + e.bq;
    // motor targets need q, qd before the velocity update; keep what is needed
    double target[ND];
#pragma unroll
    for (int i = 0; i < ND; i++) {
        double t = kArmKp * (qdes[i] - e.q[i]) / dt;
        t = t > kArmMaxVel ? kArmMaxVel : t;
        t = t < -kArmMaxVel ? -kArmMaxVel : t;
        target[i] = t;
    }
    // -- ABA: backward sweep from link 7 (frame/velocity already in R, p, w, vo), forward sweep, M^-1
    {
        ArtInertia IA;
#pragma unroll
        for (int k = 0; k < 6; k++) { IA.A[k] = 0; IA.M[k] = 0; }
#pragma unroll
        for (int k = 0; k < 9; k++) IA.H[k] = 0;
        double pAw[3] = {0, 0, 0}, pAv[3] = {0, 0, 0};
        aba_backward_step<6>(e, sc, R, p, w, vo, IA, pAw, pAv); aba_backward_step<5>(e, sc, R, p, w, vo, IA, pAw, pAv);
        aba_backward_step<4>(e, sc, R, p, w, vo, IA, pAw, pAv); aba_backward_step<3>(e, sc, R, p, w, vo, IA, pAw, pAv);
        aba_backward_step<2>(e, sc, R, p, w, vo, IA, pAw, pAv); aba_backward_step<1>(e, sc, R, p, w, vo, IA, pAw, pAv);
        aba_backward_step<0>(e, sc, R, p, w, vo, IA, pAw, pAv);
    }
    double qdd[ND], W[ND][ND];
    aba_forward_and_minv(e, sc, qdd, W);
#pragma unroll
    for (int i = 0; i < ND; i++) e.qd[i] += dt * qdd[i];
    e.bqd += dt * kGravityZ;
    if constexpr (NB == 2) e.b2qd += dt * kGravityZ;

    // -- rows.  Register rows: 7 arm motors, button motor, button limit.  LDS rows: arm limits, contacts.
    double dinv[ND];
    const double arm_bound = kArmMaxForce * dt;
#pragma unroll
    for (int i = 0; i < ND; i++) dinv[i] = 1.0 / W[i][i];
    const double wb = 1.0 / kCapMass, dinvb = 1.0 / wb;
    double rhs_bm, bound_bm, app_bm = 0.0;
    if (e.motor_on) { rhs_bm = (kButtonKp * (kButtonTarget - e.bq) / dt - e.bqd) * dinvb; bound_bm = kButtonMaxForce * dt; }
    else { rhs_bm = (0.0 - e.bqd) * dinvb; bound_bm = kDefaultMotorImpulse; }
    int ngen = 0;
#pragma unroll
    for (int i = 0; i < ND; i++) {
        const double pen_lo = e.q[i] - kJointLower[i], pen_hi = kJointUpper[i] - e.q[i];
        if (pen_lo <= kLimitActivationVel * dt || pen_hi <= kLimitActivationVel * dt) {
            double J[ND];
#pragma unroll
            for (int j = 0; j < ND; j++) J[j] = 0.0;
            if (pen_lo <= kLimitActivationVel * dt) {
                J[i] = 1.0;
                add_generic_row(sc, ngen, J, 0.0, W, pen_lo > 0 ? -pen_lo / dt : 0.0, pen_lo > 0 ? 0.0 : -pen_lo * kErp / dt, e, 0.0, 0.0, kLimitMaxImpulse);
            }
            if (pen_hi <= kLimitActivationVel * dt) {
                J[i] = -1.0;
                add_generic_row(sc, ngen, J, 0.0, W, pen_hi > 0 ? -pen_hi / dt : 0.0, pen_hi > 0 ? 0.0 : -pen_hi * kErp / dt, e, 0.0, 0.0, kLimitMaxImpulse);
            }
        }
    }
    const int nlim = ngen;
    // button limit rows (scalar, always present): lower J = +1, upper J = -1.  A stop that is still `pen`
    // away only forbids approaching it faster than pen/dt; a violated stop pushes back with erp.
    double rhs_blo, app_blo = 0.0, rhs_bhi, app_bhi = 0.0;
    {
        const double pen_lo = e.bq - kGliderLower, pen_hi = kGliderUpper - e.bq;
        rhs_blo = ((pen_lo > 0 ? -pen_lo / dt : 0.0) - e.bqd) * dinvb + (pen_lo > 0 ? 0.0 : -pen_lo * kErp / dt) * dinvb;
        rhs_bhi = ((pen_hi > 0 ? -pen_hi / dt : 0.0) + e.bqd) * dinvb + (pen_hi > 0 ? 0.0 : -pen_hi * kErp / dt) * dinvb;
    }
    // second button (NB == 2): same scalar rows on its own glider
    double rhs_bm2 = 0.0, app_bm2 = 0.0, rhs_b2lo = 0.0, app_b2lo = 0.0, rhs_b2hi = 0.0, app_b2hi = 0.0, cap2_z0 = 0.0;
    uint32_t bsel = 0;          // bit k: generic row k pushes on button 2's glider
    if constexpr (NB == 2) {
        cap2_z0 = e.bz + kGliderOriginZ + e.b2q;
        if (e.motor_on) rhs_bm2 = (kButtonKp * (kButtonTarget - e.b2q) / dt - e.b2qd) * dinvb;
        else rhs_bm2 = (0.0 - e.b2qd) * dinvb;
        const double pen_lo = e.b2q - kGliderLower, pen_hi = kGliderUpper - e.b2q;
        rhs_b2lo = ((pen_lo > 0 ? -pen_lo / dt : 0.0) - e.b2qd) * dinvb + (pen_lo > 0 ? 0.0 : -pen_lo * kErp / dt) * dinvb;
        rhs_b2hi = ((pen_hi > 0 ? -pen_hi / dt : 0.0) + e.b2qd) * dinvb + (pen_hi > 0 ? 0.0 : -pen_hi * kErp / dt) * dinvb;
        e.contact_body1 = 0; e.contact_body2 = 0;
    }
    e.contact_button = 0; e.contact_table = 0;
    // walk the kinematics again for contact Jacobians only when some sphere is close
    for (int s = 0; s < kNSphere; s++) {
        const double rad = kSphere[s][3];
        if (cc[s][2] - rad - kTableTopZ < kContactThreshold) e.contact_table = 1;
        for (int sh = 0; sh < 2 * NB; sh++) {
            double n[3];
            const int shape = sh & 1;                                   // 0 cap, 1 base
            const bool b2 = NB == 2 && sh >= 2;
            const double bx_ = b2 ? e.b2x : e.bx, by_ = b2 ? e.b2y : e.by, cz0 = b2 ? cap2_z0 : cap_z0;
            const double dist = shape == 0
                ? sphere_cylinder(cc[s], rad, bx_, by_, kCapRadius, cz0, cz0 + kCapHeight, n)
                : sphere_cylinder(cc[s], rad, bx_, by_, kBaseRadius, e.bz, e.bz + kBaseHeight, n);
            if (!(dist < kContactThreshold)) continue;
            if (shape == 0 && !b2) e.contact_button = 1;
            if constexpr (NB == 2) { if (b2) e.contact_body2 = 1; else e.contact_body1 = 1; }
            double pt[3], J[ND];
#pragma unroll
            for (int k = 0; k < 3; k++) pt[k] = cc[s][k] - rad * n[k];
#pragma unroll
            for (int j = 0; j < ND; j++) {
                double z[3], d[3], c[3];
#pragma unroll
                for (int k = 0; k < 3; k++) { z[k] = sc.at(SC_Z + 3 * j + k); d[k] = pt[k] - sc.at(SC_P + 3 * j + k); }
                cross3(z, d, c);
                J[j] = dot3(n, c);
            }
            const double allow = dist > 0 ? -dist / dt : 0.0;
            const double pos_err = dist > 0 ? 0.0 : -dist * kErp / dt;
            if (b2 && ngen < kMaxGenRows) bsel |= 1u << ngen;
            add_generic_row(sc, ngen, J, shape == 0 ? -n[2] : 0.0, W, allow, pos_err, e, b2 ? e.b2qd : e.bqd, 0.0, 1e10);
        }
    }
#ifdef SRL_ROW_STAT_HOOK          // host-side instrumentation of the test harness only (kuka_hostcheck.cpp)
    SRL_ROW_STAT_HOOK(ngen, nlim);
#endif
    // -- projected Gauss-Seidel, 150 sweeps.  Row order: arm motors, button motor, arm limits, button
    //    limits, contacts (LDS rows [0, nlim) are the arm limits, [nlim, ngen) the contacts).  Four wave-uniform
    //    code paths: 1 no generic rows, 2a <= 2 contact rows, 2b more contact rows, 3 arm-limit rows present.
    //    The arm motor block is iterated in impulse space: lam_i <- clamp((c_i - g_i - sum_{j!=i} W_ij lam_j) / W_ii),
    //    which is the same Gauss-Seidel update as accumulating dv = W lam row by row, with a 4-deep
    //    dependent chain per row and 6 FMAs instead of 7 + bookkeeping; the velocity change is formed once
    //    at the end.  The button DoF stays in velocity space (scalar dvb).
#define SRL_W(i, j) ((i) <= (j) ? W[i][j] : W[j][i])
    // Rows are pre-scaled by 1 / W_ii (Ws_ij = W_ij / W_ii, cs_i = (c_i - g_i) / W_ii) so that an arm-row update is
    //   lam_j = clamp(cur_j);  cur_i -= Ws_ij lam_j (i > j: rows still to come);  nxt_i -= Ws_ij lam_j (i < j: next sweep)
    // with cur_i = cs_i - sum_{j<i} Ws_ij lam_j(this sweep) - sum_{j>i} Ws_ij lam_j(previous sweep): the only
    // dependent chain per row is fma -> max -> min.  The three scalar button rows clamp the impulse increment
    // against (lo - applied, hi - applied), known one sweep ahead: fma -> max -> min -> fma.
    double Ws[ND][ND], lam[ND], g[ND], cs[ND], cur[ND], nxt[ND], dvb = 0.0, dvb2 = 0.0;
#pragma unroll
    for (int i = 0; i < ND; i++) {
        lam[i] = 0.0; g[i] = 0.0; cs[i] = (target[i] - e.qd[i]) * dinv[i]; cur[i] = cs[i]; nxt[i] = 0.0;
#pragma unroll
        for (int j = 0; j < ND; j++) Ws[i][j] = j == i ? 0.0 : SRL_W(i, j) * dinv[i];
    }
    const double blim = kLimitMaxImpulse;
#define SRL_BUTTON_ROW_ON(dv_, app, rhs, jsign, lo, hi)                                  \
    {                                                                                    \
        const double d__ = fmin(fmax((rhs) - (jsign) * (dv_) * dinvb, (lo) - (app)), (hi) - (app)); \
        (dv_) += (jsign) * d__ * wb;                                                     \
        (app) += d__;                                                                    \
    }
#define SRL_BUTTON_ROW(app, rhs, jsign, lo, hi) SRL_BUTTON_ROW_ON(dvb, app, rhs, jsign, lo, hi)
    // the scalar rows of button 2 only touch dvb2, so their position relative to button 1's scalar rows is immaterial
#define SRL_BUTTON2_MOTOR  if constexpr (NB == 2) { SRL_BUTTON_ROW_ON(dvb2, app_bm2, rhs_bm2, 1.0, -bound_bm, bound_bm) }
#define SRL_BUTTON2_LIMITS if constexpr (NB == 2) { SRL_BUTTON_ROW_ON(dvb2, app_b2lo, rhs_b2lo, 1.0, 0.0, blim) SRL_BUTTON_ROW_ON(dvb2, app_b2hi, rhs_b2hi, -1.0, 0.0, blim) }
    // FUSED = 1: cs is constant over the solve, so next sweep's partial sum starts from cs at its first term
#define SRL_ARM_ROWS(FUSED)                                                              \
    _Pragma("unroll") for (int j = 0; j < ND; j++) {                                     \
        const double l = fmin(fmax(cur[j], -arm_bound), arm_bound);                      \
        lam[j] = l;                                                                      \
        _Pragma("unroll") for (int i = j + 1; i < ND; i++) cur[i] -= Ws[i][j] * l;       \
        _Pragma("unroll") for (int i = 0; i < j; i++) {                                  \
            if ((FUSED) && i == j - 1) nxt[i] = cs[i] - Ws[i][j] * l;                    \
            else nxt[i] -= Ws[i][j] * l;                                                 \
        }                                                                                \
    }
    if (!SRL_ANY(ngen > 0)) {
        // -- path 1: the whole wavefront is free of limit / contact rows: one straight-line block per sweep.
        //    Without contact rows the button glider is decoupled from the arm: its three scalar rows (21 of the sweep's
        //    79 instructions) form their own little iteration on dvb.  After three sweeps it is usually PERIODIC — the
        //    sweep maps dvb onto itself bit for bit while only the applied impulses keep growing (motor pushing the cap
        //    against its stop) or nothing changes at all.  When every row of every lane is provably in a mode that
        //    the remaining sweeps cannot leave (unclamped with room for the impulse drift, or held at a zero bound
        //    with zero applied impulse), the remaining sweeps would recompute the same dvb: they run the arm rows only.
        //    Bit-identical to the full iteration by construction; any doubt falls back to it.
        int it = 0;
        bool skip_button = false;
        if (NB == 1 && kSolverIters > 3) {
            double dvb_prev = 0.0, xm = 0.0, dm = 0.0, xl = 0.0, dl = 0.0, xh = 0.0, dh = 0.0;
            for (; it < 3; it++) {
                SRL_ARM_ROWS(1)
                dvb_prev = dvb;
                xm = rhs_bm - dvb * dinvb;  dm = fmin(fmax(xm, -bound_bm - app_bm), bound_bm - app_bm); dvb += dm * wb; app_bm += dm;
                xl = rhs_blo - dvb * dinvb; dl = fmin(fmax(xl, 0.0 - app_blo), blim - app_blo);         dvb += dl * wb; app_blo += dl;
                xh = rhs_bhi + dvb * dinvb; dh = fmin(fmax(xh, 0.0 - app_bhi), blim - app_bhi);         dvb -= dh * wb; app_bhi += dh;
#pragma unroll
                for (int i = 0; i < ND - 1; i++) cur[i] = nxt[i];
                cur[ND - 1] = cs[ND - 1];
            }
            const double left = (double)(kSolverIters - 3 + 1) * (1.0 + 1e-6), room = 1.0 - 1e-6;
            const bool stable_m = dm == xm && fabs(app_bm) + left * fabs(xm) <= bound_bm * room;
            const bool stable_l = (dl == xl && xl >= 0.0 && app_blo + left * xl <= blim * room) || (dl == 0.0 && app_blo == 0.0 && xl <= 0.0);
            const bool stable_h = (dh == xh && xh >= 0.0 && app_bhi + left * xh <= blim * room) || (dh == 0.0 && app_bhi == 0.0 && xh <= 0.0);
            const bool ok = dvb == dvb_prev && stable_m && stable_l && stable_h;     // dvb_prev: value the last sweep started from
            skip_button = !SRL_ANY(!ok);
        }
        if (skip_button) {
            for (; it < kSolverIters; it++) {
                SRL_ARM_ROWS(1)
#pragma unroll
                for (int i = 0; i < ND - 1; i++) cur[i] = nxt[i];
                cur[ND - 1] = cs[ND - 1];
            }
        } else {
            for (; it < kSolverIters; it++) {
                SRL_ARM_ROWS(1)
                SRL_BUTTON_ROW(app_bm, rhs_bm, 1.0, -bound_bm, bound_bm)
                SRL_BUTTON_ROW(app_blo, rhs_blo, 1.0, 0.0, blim)
                SRL_BUTTON_ROW(app_bhi, rhs_bhi, -1.0, 0.0, blim)
                SRL_BUTTON2_MOTOR
                SRL_BUTTON2_LIMITS
#pragma unroll
                for (int i = 0; i < ND - 1; i++) cur[i] = nxt[i];
                cur[ND - 1] = cs[ND - 1];
            }
        }
    } else if (!SRL_ANY(nlim > 0) && !SRL_ANY(ngen > 2)) {
        // -- path 2a (the common contact case: at most two contact rows per lane, no arm joint near its stop).  Both rows
        //    live in VGPRs (lanes without one carry an all-zero row, a no-op) and are folded into the arm rows' scatter
        //    scheme: a contact impulse change d moves the arm rows' constant terms directly (cs_i -= d WJ_i / W_ii), so the
        //    sweep's tail is cur = cs + nxt, and a row's own residual needs WJ.lam (7 FMAs) plus the 2x2 row-row
        //    couplings K_kr = (J_k . WJ_r) / D_k instead of J.g (7 more).  g = sum_k WJ_k mu_k is formed once at the end.
        double rC[2][ND], rWJs[2][ND], rWJ[2][ND], rK[2][2], rJbD[2], rWJb[2], rRhs[2], rLo[2], rHi[2], rMu[2];
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const bool has = k < ngen;
            const int base = k * ROW_STRIDE;
            const double di = has ? sc.row(base + ROW_DINV) : 0.0;
#pragma unroll
            for (int i = 0; i < ND; i++) {
                rWJ[k][i] = has ? sc.row(base + ROW_WJ + i) : 0.0;
                rC[k][i] = rWJ[k][i] * di;
                rWJs[k][i] = rWJ[k][i] * dinv[i];
            }
            rJbD[k] = has ? sc.row(base + ROW_JB) * di : 0.0; rWJb[k] = has ? sc.row(base + ROW_WJB) : 0.0;
            rRhs[k] = has ? sc.row(base + ROW_RHS) : 0.0;
            rLo[k] = has ? sc.row(base + ROW_LO) : 0.0; rHi[k] = has ? sc.row(base + ROW_HI) : 0.0; rMu[k] = 0.0;
        }
#pragma unroll
        for (int k = 0; k < 2; k++)
#pragma unroll
            for (int r = 0; r < 2; r++) {
                double a = 0.0;
#pragma unroll
                for (int i = 0; i < ND; i++) a += ((k < ngen) ? sc.row(k * ROW_STRIDE + ROW_J + i) : 0.0) * rWJ[r][i];
                rK[k][r] = a * ((k < ngen) ? sc.row(k * ROW_STRIDE + ROW_DINV) : 0.0);
            }
        const bool rSel[2] = {NB == 2 && (bsel & 1u) != 0, NB == 2 && (bsel & 2u) != 0};
        const bool second = SRL_ANY(ngen > 1);
        for (int it = 0; it < kSolverIters; it++) {
            SRL_ARM_ROWS(0)
            SRL_BUTTON_ROW(app_bm, rhs_bm, 1.0, -bound_bm, bound_bm)
            SRL_BUTTON_ROW(app_blo, rhs_blo, 1.0, 0.0, blim)
            SRL_BUTTON_ROW(app_bhi, rhs_bhi, -1.0, 0.0, blim)
            SRL_BUTTON2_MOTOR
            SRL_BUTTON2_LIMITS
#pragma unroll
            for (int k = 0; k < 2; k++) {
                if (k == 1 && !second) break;
                double a = rRhs[k] - rJbD[k] * (rSel[k] ? dvb2 : dvb);
#pragma unroll
                for (int i = 0; i < ND; i++) a -= rC[k][i] * lam[i];
                a -= rK[k][0] * rMu[0];
                a -= rK[k][1] * rMu[1];
                const double d = fmin(fmax(a, rLo[k] - rMu[k]), rHi[k] - rMu[k]);
                rMu[k] += d;
#pragma unroll
                for (int i = 0; i < ND; i++) cs[i] -= d * rWJs[k][i];
                if (rSel[k]) dvb2 += d * rWJb[k]; else dvb += d * rWJb[k];
            }
#pragma unroll
            for (int i = 0; i < ND; i++) { cur[i] = cs[i] + nxt[i]; nxt[i] = 0.0; }
        }
#pragma unroll
        for (int i = 0; i < ND; i++) g[i] = rWJ[0][i] * rMu[0] + rWJ[1][i] * rMu[1];
    } else if (!SRL_ANY(nlim > 0)) {
        // -- path 2b: contact rows only (no arm joint near its stop), more than two on some lane.  The first two contact rows of every lane are
        //    held in VGPRs for all 150 sweeps (lanes without one carry an all-zero row, which is a no-op), the
        //    rest stay in LDS.  Arm velocity change so far = W lam + g with g = sum_k WJ_k mu_k.
        double rJ[2][ND], rWJ[2][ND], rJb[2], rWJb[2], rDinv[2], rRhs[2], rLo[2], rHi[2], rMu[2];
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const bool has = k < ngen;
            const int base = k * ROW_STRIDE;
#pragma unroll
            for (int i = 0; i < ND; i++) { rJ[k][i] = has ? sc.row(base + ROW_J + i) : 0.0; rWJ[k][i] = has ? sc.row(base + ROW_WJ + i) : 0.0; }
            rJb[k] = has ? sc.row(base + ROW_JB) : 0.0; rWJb[k] = has ? sc.row(base + ROW_WJB) : 0.0;
            rDinv[k] = has ? sc.row(base + ROW_DINV) : 0.0; rRhs[k] = has ? sc.row(base + ROW_RHS) : 0.0;
            rLo[k] = has ? sc.row(base + ROW_LO) : 0.0; rHi[k] = has ? sc.row(base + ROW_HI) : 0.0; rMu[k] = 0.0;
        }
        const bool rSel[2] = {NB == 2 && (bsel & 1u) != 0, NB == 2 && (bsel & 2u) != 0};
        const bool second = SRL_ANY(ngen > 1);
        for (int it = 0; it < kSolverIters; it++) {
            SRL_ARM_ROWS(0)
            SRL_BUTTON_ROW(app_bm, rhs_bm, 1.0, -bound_bm, bound_bm)
            SRL_BUTTON_ROW(app_blo, rhs_blo, 1.0, 0.0, blim)
            SRL_BUTTON_ROW(app_bhi, rhs_bhi, -1.0, 0.0, blim)
            SRL_BUTTON2_MOTOR
            SRL_BUTTON2_LIMITS
#pragma unroll
            for (int k = 0; k < 2; k++) {
                if (k == 1 && !second) break;
                double jdv = rJb[k] * (rSel[k] ? dvb2 : dvb);
#pragma unroll
                for (int i = 0; i < ND; i++) jdv += rWJ[k][i] * lam[i] + rJ[k][i] * g[i];
                const double d = fmin(fmax(rRhs[k] - jdv * rDinv[k], rLo[k] - rMu[k]), rHi[k] - rMu[k]);
                rMu[k] += d;
#pragma unroll
                for (int i = 0; i < ND; i++) g[i] += d * rWJ[k][i];
                if (rSel[k]) dvb2 += d * rWJb[k]; else dvb += d * rWJb[k];
            }
            for (int k = 2; SRL_ANY(k < ngen); k++)
                if (k < ngen) pgs_generic_row<NB>(sc, k, lam, g, dvb, dvb2, bsel);
#pragma unroll
            for (int i = 0; i < ND; i++) { cs[i] = ((target[i] - e.qd[i]) - g[i]) * dinv[i]; cur[i] = cs[i] + nxt[i]; nxt[i] = 0.0; }
        }
    } else {
        // -- path 3 (rare): arm-limit rows present; every generic row through LDS, in the general row order
        for (int it = 0; it < kSolverIters; it++) {
            SRL_ARM_ROWS(0)
            SRL_BUTTON_ROW(app_bm, rhs_bm, 1.0, -bound_bm, bound_bm)
            SRL_BUTTON2_MOTOR
            for (int k = 0; SRL_ANY(k < nlim); k++)
                if (k < nlim) pgs_generic_row<NB>(sc, k, lam, g, dvb, dvb2, bsel);
            SRL_BUTTON_ROW(app_blo, rhs_blo, 1.0, 0.0, blim)
            SRL_BUTTON_ROW(app_bhi, rhs_bhi, -1.0, 0.0, blim)
            SRL_BUTTON2_LIMITS
            for (int k = nlim; SRL_ANY(k < ngen); k++)
                if (k < ngen) pgs_generic_row<NB>(sc, k, lam, g, dvb, dvb2, bsel);
#pragma unroll
            for (int i = 0; i < ND; i++) { cs[i] = ((target[i] - e.qd[i]) - g[i]) * dinv[i]; cur[i] = cs[i] + nxt[i]; nxt[i] = 0.0; }
        }
    }
#undef SRL_ARM_ROWS
#undef SRL_BUTTON_ROW
#undef SRL_BUTTON_ROW_ON
#undef SRL_BUTTON2_MOTOR
#undef SRL_BUTTON2_LIMITS
    double dv[ND];
#pragma unroll
    for (int i = 0; i < ND; i++) {
        double a = g[i];
#pragma unroll
        for (int j = 0; j < ND; j++) a += SRL_W(i, j) * lam[j];
        dv[i] = a;
    }
#undef SRL_W
    // -- semi-implicit Euler, then refresh sin/cos and the gripper position
#pragma unroll
    for (int i = 0; i < ND; i++) { e.qd[i] += dv[i]; e.q[i] += dt * e.qd[i]; }
    e.bqd += dvb;
    e.bq += dt * e.bqd;
    if constexpr (NB == 2) { e.b2qd += dvb2; e.b2q += dt * e.b2qd; }
    update_trig_and_gripper(e);
}

}  // namespace kuka
}  // namespace srl
