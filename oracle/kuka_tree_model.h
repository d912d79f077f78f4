/* kuka_tree_model.h — the articulated model the Kuka oracle integrates, as ONE table that covers both variants:
 *   lumped : the 7-DoF arm chain with the gripper lumped rigidly into link_7 (rounds 1-2; constants of kuka_model.h)
 *   full   : the 12-DoF tree of pybullet_data/kuka_iiwa/kuka_with_gripper2.sdf — arm J0..J6, gripper_to_arm (joint 7, about z),
 *            base_left_finger_joint (8) -> left_base_tip_joint (10), base_right_finger_joint (11) -> right_base_tip_joint (13),
 *            the two fixed joints (9, 12: finger -> finger_base) merged into their parent links exactly (composite inertia);
 *            every joint carries the POSITION_CONTROL motor the reference commands each step (kuka.py:167-187), the collision
 *            geometry is 16 spheres attached to links 5..11 (arm wrist, flange, gripper body, fingers, finger bases, tips).
 * TEST INFRASTRUCTURE ONLY.
 *
 * In-tree (verified): motor gains / forces / targets (kuka.py:33-38,167-187), initial joint values (kuka.py:65-66).
 * Out-of-tree [UNVERIFIED-MEMORY] (pybullet_data is absent; PARITY UNPINNED): every link pose, mass, inertia and joint axis of
 * the gripper part below is recalled from the SDF file; the sphere set approximates its box / STL collision shapes.  The table
 * is runtime data: tests/golden/make_kuka_pybullet_golden.py replaces it with values read from the real files. */
#ifndef ORACLE_KUKA_TREE_MODEL_H
#define ORACLE_KUKA_TREE_MODEL_H

#include "kuka_model.h"

#define TN 12                 /* max DoFs */
#define TNS 16                /* max collision spheres */

typedef struct {
    int nd;                               /* DoFs in use: 7 (lumped) or 12 (full) */
    int parent[TN];                       /* parent DoF / link, -1 = the fixed base */
    double xyz[TN][3], Rj[TN][3][3];      /* joint frame in the parent link's frame: origin, fixed rotation (child = Rj * Rot(axis, q)) */
    double axis[TN][3];                   /* joint axis in the joint (= child link) frame, unit */
    double lower[TN], upper[TN];          /* joint limits; lower > upper = none (continuous / effectively unlimited) */
    double damping[TN];                   /* SDF <dynamics><damping> */
    double mass[TN], com[TN][3], inertia[TN][6];   /* link inertia about its COM in link axes: xx xy xz yy yz zz */
    double kp[TN], max_force[TN], max_vel[TN];     /* POSITION_CONTROL motor of the joint (velocityGain = 1 everywhere) */
    int joint_index[TN];                  /* pybullet joint index of the DoF (kuka.py's motor / joint_positions index) */
    int ee_link; double ee_point[3];      /* IK end effector: kuka_end_effector_index = 6, its inertial frame origin */
    int grip_link; double grip_point[3];  /* getArmPos(): COM of link kuka_gripper_index = 8 */
    int nsphere, sphere_link[TNS];        /* collision spheres: owning link, centre in its frame, radius, combined lateral friction */
    double sphere[TNS][4], sphere_mu[TNS];
    double table_top_z, button_base_z;
    int max_generic_rows;                 /* joint-limit + contact-normal rows kept per step (each contact normal adds one friction row in the full model) */
    int friction;                         /* 1: one friction row per contact (Bullet multibody default: a single direction from btPlaneSpace1) */
    /* Solver details of the dependency (Bullet 2.87 btMultiBodyConstraintSolver / pybullet's server defaults) that are RECALLED, not
     * read, and that the PyBullet pin decides (tests/golden/fit_kuka_pin.py): a bit mask and three scalars, runtime data like the
     * rest of the table.  Defaults = rounds 1-3 behaviour.
     *   bit 0 (TM_DETAIL_ALT_SWEEP)  non-contact rows are swept backwards on even iterations (`iteration & 1 ? j : size - 1 - j`)
     *   bit 1 (TM_DETAIL_BODY_ORDER) non-contact rows in body-creation order: button before arm (kuka_button_gym_env.py:233-238), per
     *                                body its joint-limit rows, then its motors
     *   bit 2 (TM_DETAIL_FRICTION2)  a second friction row per contact along n x t1 (SOLVER_USE_2_FRICTION_DIRECTIONS), each boxed by
     *                                mu x the normal impulse; the row budget becomes min(max_generic_rows, 4) (4 + 4 + 4 bank-B slots) */
    int solver_detail;
    double contact_erp, limit_erp;        /* error reduction of contact rows (Bullet: m_erp2) / joint-limit rows, button stops included */
    double linear_slop;                   /* contact rows see penetration = distance + linear_slop (Bullet: m_linearSlop) */
} tree_model;
#define TM_DETAIL_ALT_SWEEP 1
#define TM_DETAIL_BODY_ORDER 2
#define TM_DETAIL_FRICTION2 4
/* row budget in force: the lane group's second row bank holds 12 rows — 6 normals / limits + 6 friction rows, or 4 + 4 + 4 */
static int tm_row_budget(const tree_model *m) { const int cap = (m->solver_detail & TM_DETAIL_FRICTION2) ? 4 : 6; return m->max_generic_rows < cap ? m->max_generic_rows : cap; }

static void tm_rpy_to_mat(const double rpy[3], double R[3][3]) {
    double cr = cos(rpy[0]), sr = sin(rpy[0]), cp = cos(rpy[1]), sp = sin(rpy[1]), cy = cos(rpy[2]), sy = sin(rpy[2]);
    R[0][0] = cy * cp; R[0][1] = cy * sp * sr - sy * cr; R[0][2] = cy * sp * cr + sy * sr;
    R[1][0] = sy * cp; R[1][1] = sy * sp * sr + cy * cr; R[1][2] = sy * sp * cr - cy * sr;
    R[2][0] = -sp;     R[2][1] = cp * sr;                R[2][2] = cp * cr;
}

/* rounds 1-2 model: the tables of kuka_model.h (runtime-settable through km_set_model) as a 7-link chain */
static void tm_build_lumped(tree_model *m) {
    int i, k;
    memset(m, 0, sizeof *m);
    m->nd = 7;
    for (i = 0; i < 7; i++) {
        m->parent[i] = i - 1;
        for (k = 0; k < 3; k++) { m->xyz[i][k] = KM_JOINT_XYZ[i][k]; m->com[i][k] = KM_COM[i][k]; m->axis[i][k] = k == 2 ? 1.0 : 0.0; }
        tm_rpy_to_mat(KM_JOINT_RPY[i], m->Rj[i]);
        m->lower[i] = KM_JOINT_LOWER[i]; m->upper[i] = KM_JOINT_UPPER[i]; m->damping[i] = KM_JOINT_DAMPING;
        m->mass[i] = KM_MASS[i];
        m->inertia[i][0] = KM_INERTIA[i][0]; m->inertia[i][3] = KM_INERTIA[i][1]; m->inertia[i][5] = KM_INERTIA[i][2];
        m->kp[i] = KM_ARM_KP; m->max_force[i] = KM_ARM_MAX_FORCE; m->max_vel[i] = KM_ARM_MAX_VEL;
        m->joint_index[i] = i;
    }
    m->ee_link = 6; m->grip_link = 6;
    for (k = 0; k < 3; k++) { m->ee_point[k] = KM_EE_POINT[k]; m->grip_point[k] = KM_GRIPPER_POINT[k]; }
    m->nsphere = KM_NSPHERE;
    for (i = 0; i < KM_NSPHERE; i++) { m->sphere_link[i] = 6; for (k = 0; k < 4; k++) m->sphere[i][k] = KM_SPHERE[i][k]; m->sphere_mu[i] = 0.0; }
    m->table_top_z = KM_TABLE_TOP_Z; m->button_base_z = KM_BUTTON_BASE_Z;
    m->max_generic_rows = 6; m->friction = 0;
    m->solver_detail = 0; m->contact_erp = KM_ERP; m->limit_erp = KM_ERP; m->linear_slop = 0.0;
}

/* ---- kuka_with_gripper2.sdf, gripper part [UNVERIFIED-MEMORY]: link poses in the model frame at q = 0 (xyz, rpy), inertial
 * frame offset in the link, mass, isotropic-diagonal inertia as written in the file */
typedef struct { double pose[6], ipos[3], mass, inertia[3]; } tm_sdf_link;
static const tm_sdf_link TM_LINK7       = {{0, 0, 1.261, 0, 0, 0}, {0, 0, 0.02}, 0.3, {0.001, 0.001, 0.001}};      /* lbr_iiwa_link_7 alone */
static const tm_sdf_link TM_BASE_LINK   = {{0, 0, 1.305, 0, 0, 0}, {0, 0, 0}, 1.2, {1.0, 1.0, 1.0}};               /* gripper body, visual box 0.05 0.05 0.1 */
static const tm_sdf_link TM_L_FINGER    = {{0, 0.024, 1.35, 0, -0.05, 0}, {0, 0, 0.04}, 0.2, {0.1, 0.1, 0.1}};     /* box 0.01 0.01 0.08 */
static const tm_sdf_link TM_L_FBASE     = {{-0.005, 0.024, 1.43, 0, -0.3, 0}, {-0.003, 0, 0.04}, 0.2, {0.1, 0.1, 0.1}};   /* finger_base_left.stl, fixed to left_finger */
static const tm_sdf_link TM_L_TIP       = {{-0.02, 0.024, 1.49, 0, 0.2, 0}, {-0.005, 0, 0.026}, 0.2, {0.1, 0.1, 0.1}};    /* finger_tip_left.stl */
static const tm_sdf_link TM_R_FINGER    = {{0, -0.024, 1.35, 0, 0.05, 0}, {0, 0, 0.04}, 0.2, {0.1, 0.1, 0.1}};
static const tm_sdf_link TM_R_FBASE     = {{0.005, -0.024, 1.43, 0, 0.3, 0}, {0.003, 0, 0.04}, 0.2, {0.1, 0.1, 0.1}};
static const tm_sdf_link TM_R_TIP       = {{0.02, -0.024, 1.49, 0, -0.2, 0}, {0.005, 0, 0.026}, 0.2, {0.1, 0.1, 0.1}};
/* motor forces, kuka.py:33-36: max_force 200, fingerA 2, fingerB 2.5, finger_tip 2; pybullet default positionGain 0.1 */
#define TM_GRIPPER_KP 0.1
#define TM_FINGER_A_FORCE 2.0
#define TM_FINGER_B_FORCE 2.5
#define TM_FINGER_TIP_FORCE 2.0
#define TM_MU_DEFAULT 0.25              /* 0.5 (pybullet default lateral friction) x 0.5 (button / table) */
#define TM_MU_FINGER 0.4                /* finger base / tip links: <lateral_friction>0.8 x 0.5 */

static void tm_mat_mul(const double A[3][3], const double B[3][3], double C[3][3]) {
    int i, j, k; for (i = 0; i < 3; i++) for (j = 0; j < 3; j++) { double s = 0; for (k = 0; k < 3; k++) s += A[i][k] * B[k][j]; C[i][j] = s; }
}
/* child link pose relative to the parent link (both given in the model frame): xyz, R */
static void tm_relative(const double parent_pose[6], const double child_pose[6], double xyz[3], double R[3][3]) {
    double Rp[3][3], Rc[3][3], d[3]; int i, j, k;
    tm_rpy_to_mat(parent_pose + 3, Rp); tm_rpy_to_mat(child_pose + 3, Rc);
    for (k = 0; k < 3; k++) d[k] = child_pose[k] - parent_pose[k];
    for (i = 0; i < 3; i++) { xyz[i] = Rp[0][i] * d[0] + Rp[1][i] * d[1] + Rp[2][i] * d[2]; }
    for (i = 0; i < 3; i++) for (j = 0; j < 3; j++) { double s = 0; for (k = 0; k < 3; k++) s += Rp[k][i] * Rc[k][j]; R[i][j] = s; }
}
/* inertial parameters of link dof <- one SDF link, optionally with a rigidly attached second link (fixed joint) merged in */
static void tm_set_body(tree_model *m, int dof, const tm_sdf_link *a, const tm_sdf_link *b) {
    double mt = a->mass, c[3], I[3][3] = {{0}}; int i, j, k;
    for (k = 0; k < 3; k++) c[k] = a->ipos[k];
    if (!b) {
        m->mass[dof] = a->mass;
        for (k = 0; k < 3; k++) m->com[dof][k] = a->ipos[k];
        m->inertia[dof][0] = a->inertia[0]; m->inertia[dof][3] = a->inertia[1]; m->inertia[dof][5] = a->inertia[2];
        m->inertia[dof][1] = m->inertia[dof][2] = m->inertia[dof][4] = 0.0;
        return;
    }
    {
        double xyz[3], R[3][3], cb[3], Ib[3][3], T[3][3], Rt[3][3], D[3][3] = {{0}};
        tm_relative(a->pose, b->pose, xyz, R);
        for (i = 0; i < 3; i++) cb[i] = xyz[i] + R[i][0] * b->ipos[0] + R[i][1] * b->ipos[1] + R[i][2] * b->ipos[2];
        mt = a->mass + b->mass;
        for (k = 0; k < 3; k++) c[k] = (a->mass * a->ipos[k] + b->mass * cb[k]) / mt;
        D[0][0] = b->inertia[0]; D[1][1] = b->inertia[1]; D[2][2] = b->inertia[2];
        for (i = 0; i < 3; i++) for (j = 0; j < 3; j++) Rt[i][j] = R[j][i];
        tm_mat_mul(R, D, T); tm_mat_mul(T, Rt, Ib);
        for (i = 0; i < 3; i++) I[i][i] = a->inertia[i];
        for (i = 0; i < 3; i++) for (j = 0; j < 3; j++) I[i][j] += Ib[i][j];
        {   /* parallel-axis terms of both parts about the composite COM */
            const double *cs[2]; double ms[2]; int s;
            cs[0] = a->ipos; cs[1] = cb; ms[0] = a->mass; ms[1] = b->mass;
            for (s = 0; s < 2; s++) {
                double r[3], rr; for (k = 0; k < 3; k++) r[k] = cs[s][k] - c[k];
                rr = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
                for (i = 0; i < 3; i++) for (j = 0; j < 3; j++) I[i][j] += ms[s] * ((i == j ? rr : 0.0) - r[i] * r[j]);
            }
        }
    }
    m->mass[dof] = mt;
    for (k = 0; k < 3; k++) m->com[dof][k] = c[k];
    m->inertia[dof][0] = I[0][0]; m->inertia[dof][1] = I[0][1]; m->inertia[dof][2] = I[0][2];
    m->inertia[dof][3] = I[1][1]; m->inertia[dof][4] = I[1][2]; m->inertia[dof][5] = I[2][2];
}
static void tm_add_sphere(tree_model *m, int link, const double *frame_in_link_xyz, const double frame_in_link_R[3][3], double x, double y, double z,
                          double r, double mu) {
    int s = m->nsphere++, i; const double c[3] = {x, y, z};
    m->sphere_link[s] = link; m->sphere[s][3] = r; m->sphere_mu[s] = mu;
    for (i = 0; i < 3; i++)
        m->sphere[s][i] = frame_in_link_xyz ? frame_in_link_xyz[i] + frame_in_link_R[i][0] * c[0] + frame_in_link_R[i][1] * c[1] + frame_in_link_R[i][2] * c[2] : c[i];
}

static void tm_build_full(tree_model *m) {
    static const int jidx[TN] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 11, 13};
    const tm_sdf_link *fing[2] = {&TM_L_FINGER, &TM_R_FINGER}, *fbase[2] = {&TM_L_FBASE, &TM_R_FBASE}, *tip[2] = {&TM_L_TIP, &TM_R_TIP};
    int i, k, side;
    tm_build_lumped(m);                                 /* arm joints 0..6 (frames, limits, damping, motors) */
    m->nd = 12; m->nsphere = 0; m->max_generic_rows = 6; m->friction = 1;
    for (i = 0; i < TN; i++) m->joint_index[i] = jidx[i];
    tm_set_body(m, 6, &TM_LINK7, NULL);                 /* link_7 without the lumped gripper */
    /* DoF 7: gripper_to_arm (continuous, axis z), child base_link */
    m->parent[7] = 6; tm_relative(TM_LINK7.pose, TM_BASE_LINK.pose, m->xyz[7], m->Rj[7]);
    m->axis[7][0] = 0; m->axis[7][1] = 0; m->axis[7][2] = 1;
    tm_set_body(m, 7, &TM_BASE_LINK, NULL);
    m->kp[7] = TM_GRIPPER_KP; m->max_force[7] = KM_ARM_MAX_FORCE;
    for (side = 0; side < 2; side++) {
        const int f = 8 + 2 * side, t = f + 1;          /* finger (+ fixed finger_base), tip */
        m->parent[f] = 7; tm_relative(TM_BASE_LINK.pose, fing[side]->pose, m->xyz[f], m->Rj[f]);
        m->parent[t] = f; tm_relative(fing[side]->pose, tip[side]->pose, m->xyz[t], m->Rj[t]);
        for (k = 0; k < 3; k++) { m->axis[f][k] = k == 1 ? 1.0 : 0.0; m->axis[t][k] = k == 1 ? 1.0 : 0.0; }
        tm_set_body(m, f, fing[side], fbase[side]);
        tm_set_body(m, t, tip[side], NULL);
        m->kp[f] = TM_GRIPPER_KP; m->kp[t] = TM_GRIPPER_KP;
        m->max_force[f] = side == 0 ? TM_FINGER_A_FORCE : TM_FINGER_B_FORCE; m->max_force[t] = TM_FINGER_TIP_FORCE;
    }
    for (i = 7; i < TN; i++) { m->lower[i] = 1.0; m->upper[i] = -1.0; m->damping[i] = 0.0; m->max_vel[i] = 1e30; }   /* limits of +-10 rad never act */
    m->ee_link = 6; m->grip_link = 8;
    for (k = 0; k < 3; k++) { m->ee_point[k] = TM_LINK7.ipos[k]; m->grip_point[k] = TM_L_FINGER.ipos[k]; }
    /* collision spheres, ordered by link */
    tm_add_sphere(m, 5, NULL, NULL, 0, 0, 0, 0.07, TM_MU_DEFAULT);             /* wrist (lbr_iiwa_link_6) */
    tm_add_sphere(m, 6, NULL, NULL, 0, 0, 0.02, 0.05, TM_MU_DEFAULT);          /* flange */
    tm_add_sphere(m, 7, NULL, NULL, 0, 0, -0.025, 0.035, TM_MU_DEFAULT);       /* gripper body box, two spheres */
    tm_add_sphere(m, 7, NULL, NULL, 0, 0, 0.025, 0.035, TM_MU_DEFAULT);
    for (side = 0; side < 2; side++) {
        const int f = 8 + 2 * side, t = f + 1; double xyz[3], R[3][3];
        tm_add_sphere(m, f, NULL, NULL, 0, 0, 0.02, 0.008, TM_MU_DEFAULT);     /* finger box */
        tm_add_sphere(m, f, NULL, NULL, 0, 0, 0.06, 0.008, TM_MU_DEFAULT);
        tm_relative(fing[side]->pose, fbase[side]->pose, xyz, R);              /* finger_base mesh, in the finger's frame */
        tm_add_sphere(m, f, xyz, R, 0, 0, 0.015, 0.012, TM_MU_FINGER);
        tm_add_sphere(m, f, xyz, R, 0, 0, 0.045, 0.012, TM_MU_FINGER);
        tm_add_sphere(m, t, NULL, NULL, 0, 0, 0.012, 0.010, TM_MU_FINGER);     /* tip mesh */
        tm_add_sphere(m, t, NULL, NULL, 0, 0, 0.032, 0.010, TM_MU_FINGER);
    }
}

/* flat float64 image of a tree_model (the layout of include/srlhip.h `srlhip_kuka_tree_model`), ints stored as doubles */
#define TM_DOUBLES (1 + TN * (1 + 3 + 9 + 3 + 2 + 1 + 1 + 3 + 6 + 3 + 1) + (1 + 3) + (1 + 3) + 1 + TNS * (1 + 4 + 1) + 2 + 2 + 4)
static void tm_to_table(const tree_model *m, double *t) {
    int k = 0, i, j;
    t[k++] = m->nd;
    for (i = 0; i < TN; i++) {
        t[k++] = m->parent[i];
        for (j = 0; j < 3; j++) t[k++] = m->xyz[i][j];
        for (j = 0; j < 9; j++) t[k++] = m->Rj[i][j / 3][j % 3];
        for (j = 0; j < 3; j++) t[k++] = m->axis[i][j];
        t[k++] = m->lower[i]; t[k++] = m->upper[i]; t[k++] = m->damping[i]; t[k++] = m->mass[i];
        for (j = 0; j < 3; j++) t[k++] = m->com[i][j];
        for (j = 0; j < 6; j++) t[k++] = m->inertia[i][j];
        t[k++] = m->kp[i]; t[k++] = m->max_force[i]; t[k++] = m->max_vel[i];
        t[k++] = m->joint_index[i];
    }
    t[k++] = m->ee_link; for (j = 0; j < 3; j++) t[k++] = m->ee_point[j];
    t[k++] = m->grip_link; for (j = 0; j < 3; j++) t[k++] = m->grip_point[j];
    t[k++] = m->nsphere;
    for (i = 0; i < TNS; i++) { t[k++] = m->sphere_link[i]; for (j = 0; j < 4; j++) t[k++] = m->sphere[i][j]; t[k++] = m->sphere_mu[i]; }
    t[k++] = m->table_top_z; t[k++] = m->button_base_z; t[k++] = m->max_generic_rows; t[k++] = m->friction;
    t[k++] = m->solver_detail; t[k++] = m->contact_erp; t[k++] = m->limit_erp; t[k++] = m->linear_slop;
}
static void tm_from_table(tree_model *m, const double *t) {
    int k = 0, i, j;
    m->nd = (int)t[k++];
    for (i = 0; i < TN; i++) {
        m->parent[i] = (int)t[k++];
        for (j = 0; j < 3; j++) m->xyz[i][j] = t[k++];
        for (j = 0; j < 9; j++) m->Rj[i][j / 3][j % 3] = t[k++];
        for (j = 0; j < 3; j++) m->axis[i][j] = t[k++];
        m->lower[i] = t[k++]; m->upper[i] = t[k++]; m->damping[i] = t[k++]; m->mass[i] = t[k++];
        for (j = 0; j < 3; j++) m->com[i][j] = t[k++];
        for (j = 0; j < 6; j++) m->inertia[i][j] = t[k++];
        m->kp[i] = t[k++]; m->max_force[i] = t[k++]; m->max_vel[i] = t[k++];
        m->joint_index[i] = (int)t[k++];
    }
    m->ee_link = (int)t[k++]; for (j = 0; j < 3; j++) m->ee_point[j] = t[k++];
    m->grip_link = (int)t[k++]; for (j = 0; j < 3; j++) m->grip_point[j] = t[k++];
    m->nsphere = (int)t[k++];
    for (i = 0; i < TNS; i++) { m->sphere_link[i] = (int)t[k++]; for (j = 0; j < 4; j++) m->sphere[i][j] = t[k++]; m->sphere_mu[i] = t[k++]; }
    m->table_top_z = t[k++]; m->button_base_z = t[k++]; m->max_generic_rows = (int)t[k++]; m->friction = (int)t[k++];
    m->solver_detail = (int)t[k++]; m->contact_erp = t[k++]; m->limit_erp = t[k++]; m->linear_slop = t[k++];
}
#endif
