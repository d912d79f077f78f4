/* kuka_model.h — constants of the Kuka-button model the oracle integrates.
 * TEST INFRASTRUCTURE ONLY.
 *
 * In-tree (verified) sources:
 *   /root/reference/environments/kuka_gym/kuka.py:21-53,63-73,167-187
 *   /root/reference/environments/kuka_gym/kuka_button_gym_env.py:17-35,219-236,347
 *   /root/reference/urdf/simple_button.urdf (joint origins, limits, masses) + mesh extents
 * Out-of-tree (RECALLED, unverifiable here — PARITY UNPINNED): link frames,
 * masses and inertias of pybullet_data/kuka_iiwa/kuka_with_gripper2.sdf
 * (pybullet==1.8.6, environment.yml:109); SURVEY.md App. B.4.  The WSG50-like
 * gripper (joints 7-13, motor-held at ~0 by kuka.py:177-187) is lumped rigidly
 * into link_7; its collision geometry is six spheres fixed in the link_7 frame. */
#ifndef ORACLE_KUKA_MODEL_H
#define ORACLE_KUKA_MODEL_H

#define KM_PI 3.14159265358979323846

#define KM_NDOF 7
#define KM_DT (1.0 / 240.0)                 /* kuka_button_gym_env.py:86  */
#define KM_SOLVER_ITERS 150                 /* :219 numSolverIterations   */
#define KM_GRAVITY_Z (-10.0)                /* :236                       */

static const double KM_BASE_POS[3] = {-0.1, 0.0, -0.15};        /* kuka.py:63 */
/* joint frame in parent link frame: xyz then URDF rpy (R = Rz(y) Ry(p) Rx(r)) */
static double KM_JOINT_XYZ[7][3] = {
    {0, 0, 0.1575}, {0, 0, 0.2025}, {0, 0.2045, 0}, {0, 0, 0.2155}, {0, 0.1845, 0}, {0, 0, 0.2155}, {0, 0.081, 0}};
static double KM_JOINT_RPY[7][3] = {
    {0, 0, 0}, {KM_PI / 2, 0, KM_PI}, {KM_PI / 2, 0, KM_PI}, {KM_PI / 2, 0, 0}, {-KM_PI / 2, KM_PI, 0},
    {KM_PI / 2, 0, 0}, {-KM_PI / 2, KM_PI, 0}};
static double KM_JOINT_LOWER[7] = {-2.96705972839, -2.09439510239, -2.96705972839, -2.09439510239,
                                         -2.96705972839, -2.09439510239, -3.05432619099};
static double KM_JOINT_UPPER[7] = {2.96705972839, 2.09439510239, 2.96705972839, 2.09439510239,
                                         2.96705972839, 2.09439510239, 3.05432619099};
static double KM_JOINT_DAMPING = 0.5;
/* inertial parameters in the link frame; link 6 = link_7 + lumped gripper */
static double KM_MASS[7] = {4.0, 4.0, 3.0, 2.7, 1.7, 1.8, 1.8};
static double KM_COM[7][3] = {{0, -0.03, 0.12}, {0.0003, 0.059, 0.042}, {0, 0.03, 0.13}, {0, 0.067, 0.034},
                                    {0.0001, 0.021, 0.076}, {0, 0.0006, 0.0004}, {0, 0, 0.31 / 3.0}};
static double KM_INERTIA[7][3] = {{0.1, 0.09, 0.02}, {0.05, 0.018, 0.044}, {0.08, 0.075, 0.01},
                                        {0.03, 0.01, 0.029}, {0.02, 0.018, 0.005}, {0.005, 0.0036, 0.0047},
                                        {0.0075, 0.0075, 0.003}};
/* initial joint state, kuka.py:65-66 (first 7 of 14) and the full constant list
 * returned by the `joints` observation (kuka_button_gym_env.py:183 quirk)      */
static const double KM_JOINT_POSITIONS[14] = {0.006418, 0.113184, -0.011401, -1.289317, 0.005379, 1.737684, -0.006539,
                                              0.000048, -0.299912, 0.000000, -0.000043, 0.299960, 0.000000, -0.000200};
static const double KM_EE_INIT[3] = {0.537, 0.0, 0.5};          /* kuka.py:73 */
/* workspace clip box, kuka.py:46-53: [small constraints][min/max][xyz] */
static const double KM_EE_BOX[2][2][3] = {{{0.35, -0.30, 0.0}, {0.65, 0.30, 0.5}},     /* random_target */
                                          {{0.50, -0.17, 0.0}, {0.65, 0.22, 0.5}}};    /* fixed target  */
/* arm motors, kuka.py:167-170; impulse bound = force * dt */
#define KM_ARM_KP 0.3
#define KM_ARM_MAX_VEL 0.35
#define KM_ARM_MAX_FORCE 200.0
/* points fixed in the link_7 frame */
static double KM_EE_POINT[3] = {0, 0, 0.02};            /* IK end effector = link_7 inertial frame   */
static double KM_GRIPPER_POINT[3] = {0, 0.024, 0.10};   /* COM of gripper link 8 (getArmPos)         */
#define KM_NSPHERE 6
static double KM_SPHERE[KM_NSPHERE][4] = {              /* centre xyz, radius                        */
    {0, 0.020, 0.255, 0.015}, {0, -0.020, 0.255, 0.015},      /* finger tips                                */
    {0, 0.035, 0.200, 0.020}, {0, -0.035, 0.200, 0.020},      /* finger bodies                              */
    {0, 0, 0.100, 0.060}, {0, 0, 0.0, 0.070}};                /* gripper body, wrist                        */
#define KM_IK_DAMPING 1e-5                                    /* kuka.py:41-42 jd                           */
#define KM_IK_DAMPING_DEFAULT 0.5                             /* pybullet server default when no jointDamping is passed (Kuka2Button) [UNVERIFIED-MEMORY] */
#define KM_IK_CROSS_DET 3e-9                                  /* det(J^T J + damping I) below this: the controller's closed loop amplifies float64 noise (kuka_oracle.c ik_conditioning_note; srlhip.h SRLHIP_KUKA_IK_CROSS_DET) */
#define KM_BUTTON1_Y_2B 0.125                                 /* kuka_2button_gym_env.py:56-62 */
#define KM_BUTTON2_Y_2B (-0.125)                              /* :66-70 */
#define KM_Z_TABLE (-0.2)                                     /* kuka_button_gym_env.py Z_TABLE */
#define KM_MAX_STEPS_2BUTTON 1500                             /* kuka_2button_gym_env.py:3 */
#define KM_IK_MAX_ANGLE (45.0 * KM_PI / 180.0)                /* BussIK MaxAngleDLS                          */

/* button (urdf/simple_button.urdf): static base, prismatic cap ("glider") */
static double KM_TABLE_TOP_Z = -0.195;   /* table.urdf top box, base at z=-0.82 (recalled)                     */
static double KM_BUTTON_BASE_Z = -0.195; /* spawned at Z_TABLE=-0.2 inside the table top, settles on it        */
#define KM_BUTTON_X 0.5
#define KM_BUTTON_Y 0.0
#define KM_GLIDER_ORIGIN_Z 0.005     /* simple_button.urdf:13 */
#define KM_GLIDER_LOWER 0.0
#define KM_GLIDER_UPPER 0.01         /* :15 */
#define KM_CAP_MASS 0.1              /* :63 */
#define KM_CAP_RADIUS 0.09           /* button.dae extents */
#define KM_CAP_HEIGHT 0.03
#define KM_BASE_RADIUS 0.10          /* base_button.dae + base_cylinder.dae */
#define KM_BASE_HEIGHT 0.03
#define KM_BUTTON_DISTANCE_HEIGHT 0.28   /* kuka_button_gym_env.py:35 */
#define KM_BUTTON_TARGET 0.1             /* :347 targetPosition */
#define KM_BUTTON_KP 0.1                 /* pybullet default positionGain */
#define KM_BUTTON_MAX_FORCE 100000.0     /* pybullet default force (recalled) */
#define KM_DEFAULT_MOTOR_IMPULSE 1.0     /* pybullet createJointMotors default velocity motor (recalled) */
#define KM_LIMIT_MAX_IMPULSE 100.0       /* btMultiBodyConstraint default m_maxAppliedImpulse */
#define KM_LIMIT_ACTIVATION_VEL 10.0    /* limit rows exist when stop distance / dt <= this (rad/s) */
#define KM_ERP 0.2                       /* btContactSolverInfo m_erp / m_erp2 defaults */
#define KM_CONTACT_THRESHOLD 0.002       /* manifold points live below this separation */

/* env wrapper constants, kuka_button_gym_env.py:17-35 */
#define KM_MAX_STEPS 1000
#define KM_N_CONTACTS_BEFORE_TERMINATION 5
#define KM_N_STEPS_OUTSIDE_SAFETY_SPHERE 5000
#define KM_DELTA_V 0.03
#define KM_DELTA_V_CONTINUOUS 0.0035
#define KM_DELTA_THETA 0.1
#define KM_NOISE_STD 0.01
#define KM_NOISE_STD_CONTINUOUS 0.0001
#define KM_NOISE_STD_JOINTS 0.002
#define KM_N_RANDOM_ACTIONS_AT_INIT 5
#define KM_N_SETTLE_STEPS 500

/* The RECALLED part of the model (everything above that is not pinned by the in-tree reference source) as one runtime table,
 * in the layout of `srlhip_kuka_model` (include/srlhip.h): joint_xyz[7][3] joint_rpy[7][3] joint_lower[7] joint_upper[7]
 * joint_damping mass[7] com[7][3] inertia[7][3] ee_point[3] gripper_point[3] sphere[6][4] table_top_z button_base_z.
 * tests/golden/make_kuka_pybullet_golden.py extracts it from pybullet_data when PyBullet is available; km_set_model()
 * installs it in the including translation unit's copies (kuka_oracle.c and raster_oracle.c each export a setter). */
#define KM_MODEL_DOUBLES 138
static void km_get_model(double *t) {
    int k = 0, i, j;
    for (i = 0; i < 7; i++) for (j = 0; j < 3; j++) t[k++] = KM_JOINT_XYZ[i][j];
    for (i = 0; i < 7; i++) for (j = 0; j < 3; j++) t[k++] = KM_JOINT_RPY[i][j];
    for (i = 0; i < 7; i++) t[k++] = KM_JOINT_LOWER[i];
    for (i = 0; i < 7; i++) t[k++] = KM_JOINT_UPPER[i];
    t[k++] = KM_JOINT_DAMPING;
    for (i = 0; i < 7; i++) t[k++] = KM_MASS[i];
    for (i = 0; i < 7; i++) for (j = 0; j < 3; j++) t[k++] = KM_COM[i][j];
    for (i = 0; i < 7; i++) for (j = 0; j < 3; j++) t[k++] = KM_INERTIA[i][j];
    for (j = 0; j < 3; j++) t[k++] = KM_EE_POINT[j];
    for (j = 0; j < 3; j++) t[k++] = KM_GRIPPER_POINT[j];
    for (i = 0; i < KM_NSPHERE; i++) for (j = 0; j < 4; j++) t[k++] = KM_SPHERE[i][j];
    t[k++] = KM_TABLE_TOP_Z; t[k++] = KM_BUTTON_BASE_Z;
}
static void km_set_model(const double *t) {
    int k = 0, i, j;
    for (i = 0; i < 7; i++) for (j = 0; j < 3; j++) KM_JOINT_XYZ[i][j] = t[k++];
    for (i = 0; i < 7; i++) for (j = 0; j < 3; j++) KM_JOINT_RPY[i][j] = t[k++];
    for (i = 0; i < 7; i++) KM_JOINT_LOWER[i] = t[k++];
    for (i = 0; i < 7; i++) KM_JOINT_UPPER[i] = t[k++];
    KM_JOINT_DAMPING = t[k++];
    for (i = 0; i < 7; i++) KM_MASS[i] = t[k++];
    for (i = 0; i < 7; i++) for (j = 0; j < 3; j++) KM_COM[i][j] = t[k++];
    for (i = 0; i < 7; i++) for (j = 0; j < 3; j++) KM_INERTIA[i][j] = t[k++];
    for (j = 0; j < 3; j++) KM_EE_POINT[j] = t[k++];
    for (j = 0; j < 3; j++) KM_GRIPPER_POINT[j] = t[k++];
    for (i = 0; i < KM_NSPHERE; i++) for (j = 0; j < 4; j++) KM_SPHERE[i][j] = t[k++];
    KM_TABLE_TOP_Z = t[k++]; KM_BUTTON_BASE_Z = t[k++];
}
#endif
