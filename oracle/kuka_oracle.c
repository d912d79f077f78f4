/* kuka_oracle.c — plain-C float64 restatement of KukaButtonGymEnv stepping.
 * TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py cpu_baseline).
 *
 * In-tree logic restated (verified line by line against the reference source, and
 * pinned by tests/golden/kuka_wrapper_reference.npz — the reference's own wrapper
 * code driven by a scripted fake pybullet):
 *   Kuka.applyAction            /root/reference/environments/kuka_gym/kuka.py:118-187
 *   KukaButtonGymEnv.reset      .../kuka_button_gym_env.py:214-281
 *   KukaButtonGymEnv.step       :293-340      step2 :342-368
 *   _termination :422-426       _reward :428-463     getSRLState :175-189
 *   Kuka2ButtonGymEnv           .../kuka_2button_gym_env.py:33-200 (reset draws, two buttons, goal switching, reward)
 *   KukaMovingButtonGymEnv      .../kuka_moving_button_gym_env.py
 *   KukaRandButtonGymEnv        .../kuka_rand_button_gym_env.py:38-127: the 20 extra np_random draws of reset() and the
 *                               placement rule of the ten distractor objects.  The objects (duck / lego / cube meshes,
 *                               types drawn from the GLOBAL unseeded np.random) and the pushed ball are free rigid bodies
 *                               inside pybullet; here they are scenery (recorded for the renderer, no dynamics).
 * Out-of-tree arithmetic restated from the dependency's PUBLISHED algorithm
 * (pybullet==1.8.6 / Bullet 2.87, absent here -> PARITY UNPINNED for this part):
 *   p.calculateInverseKinematics  one damped-least-squares step (BussIK DLS)
 *   p.stepSimulation              btMultiBodyDynamicsWorld: Featherstone ABA + joint
 *                                 motor / limit / contact rows + 150-iteration
 *                                 projected Gauss-Seidel + semi-implicit Euler
 *   p.getContactPoints            analytic sphere/cylinder/plane predicate
 * Written for readability, not speed: dense 6x6 spatial algebra, no structure exploited,
 * M^-1 obtained by seven extra ABA calls.  Compile with -ffp-contract=off. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "kuka_model.h"
#include "kuka_tree_model.h"
#include "np_random.h"
#include "philox.h"

#define N KM_NDOF              /* arm joints (the IK / command surface) */
#define RB_N 11                 /* KukaRandButton free bodies: ten distractors + the ball */
#define MAX_ROWS 96             /* 15 bank rows + 6 generic + 12 friction + (RandButton) 11 table normals + 22 table friction rows, with slack */

/* optional trace of the commands handed to the arm (wrapper pinning tests) */
static double *g_trace_ee = NULL, *g_trace_jt = NULL;
static int g_trace_n = 0;
/* Solver details of Bullet's btMultiBodyConstraintSolver that are recalled, not read (the source is absent here): they live in the
 * model table (kuka_tree_model.h: solver_detail bits, contact_erp, limit_erp, linear_slop) so that the PyBullet pin can decide them
 * as DATA; the product's tree kernel takes the same table.  kuka_oracle_set_detail() edits the bit mask of the table in use (a
 * kuka_oracle_set_full() / set_model() call rebuilds the table and clears it). */
#pragma omp threadprivate(g_trace_ee, g_trace_jt, g_trace_n)

/* ------------------------------------------------------------------ small algebra */
typedef double mat3[3][3];
typedef double vec6[6];
typedef double mat6[6][6];

static void rpy_to_mat(const double rpy[3], mat3 R) {
    double cr = cos(rpy[0]), sr = sin(rpy[0]), cp = cos(rpy[1]), sp = sin(rpy[1]), cy = cos(rpy[2]), sy = sin(rpy[2]);
    R[0][0] = cy * cp; R[0][1] = cy * sp * sr - sy * cr; R[0][2] = cy * sp * cr + sy * sr;
    R[1][0] = sy * cp; R[1][1] = sy * sp * sr + cy * cr; R[1][2] = sy * sp * cr - cy * sr;
    R[2][0] = -sp;     R[2][1] = cp * sr;                R[2][2] = cp * cr;
}
static void mat3_mul(const mat3 A, const mat3 B, mat3 C) {
    int i, j, k;
    for (i = 0; i < 3; i++) for (j = 0; j < 3; j++) { double s = 0; for (k = 0; k < 3; k++) s += A[i][k] * B[k][j]; C[i][j] = s; }
}
static void mat3_vec(const mat3 A, const double v[3], double o[3]) {
    int i; for (i = 0; i < 3; i++) o[i] = A[i][0] * v[0] + A[i][1] * v[1] + A[i][2] * v[2];
}
static void cross(const double a[3], const double b[3], double o[3]) {
    o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}
static void skew(const double v[3], mat3 S) {
    S[0][0] = 0; S[0][1] = -v[2]; S[0][2] = v[1]; S[1][0] = v[2]; S[1][1] = 0; S[1][2] = -v[0];
    S[2][0] = -v[1]; S[2][1] = v[0]; S[2][2] = 0;
}
static void mat6_vec(const mat6 A, const vec6 v, vec6 o) {
    int i, k; for (i = 0; i < 6; i++) { double s = 0; for (k = 0; k < 6; k++) s += A[i][k] * v[k]; o[i] = s; }
}
static void mat6T_vec(const mat6 A, const vec6 v, vec6 o) {
    int i, k; for (i = 0; i < 6; i++) { double s = 0; for (k = 0; k < 6; k++) s += A[k][i] * v[k]; o[i] = s; }
}
/* spatial cross products: crm(v) m  and  crf(v) f = -crm(v)^T f */
static void crm_vec(const vec6 v, const vec6 m, vec6 o) {
    double a[3], b[3], c[3];
    cross(v, m, a); o[0] = a[0]; o[1] = a[1]; o[2] = a[2];
    cross(v + 3, m, b); cross(v, m + 3, c);
    o[3] = b[0] + c[0]; o[4] = b[1] + c[1]; o[5] = b[2] + c[2];
}
static void crf_vec(const vec6 v, const vec6 f, vec6 o) {
    double a[3], b[3], c[3];
    cross(v, f, a); cross(v + 3, f + 3, b);
    o[0] = a[0] + b[0]; o[1] = a[1] + b[1]; o[2] = a[2] + b[2];
    cross(v, f + 3, c); o[3] = c[0]; o[4] = c[1]; o[5] = c[2];
}

/* ------------------------------------------------------------------ the model in use */
/* One table for both variants (kuka_tree_model.h): g_m.nd = 7 (gripper lumped into link_7, rounds 1-2) or 12 (full gripper tree).
 * kuka_oracle_set_full() switches; tests are single-threaded callers (OpenMP workers only read the table). */
static tree_model g_m;
static int g_m_ready = 0, g_full = 0;
static void inertia_refresh(void);
static void model_refresh(void) { if (g_full) tm_build_full(&g_m); else tm_build_lumped(&g_m); g_m_ready = 1; inertia_refresh(); }
static const tree_model *model(void) { if (!g_m_ready) model_refresh(); return &g_m; }
#define ND (model()->nd)
void kuka_oracle_set_detail(int mask) { (void)model(); g_m.solver_detail = mask; }
int kuka_oracle_get_detail(void) { return model()->solver_detail; }

/* ------------------------------------------------------------------ kinematics */
/* rotation about a unit axis (Rodrigues) */
static void axis_rotation(const double a[3], double q, mat3 R) {
    double c = cos(q), s = sin(q), v = 1.0 - c;
    R[0][0] = c + a[0] * a[0] * v;        R[0][1] = a[0] * a[1] * v - a[2] * s; R[0][2] = a[0] * a[2] * v + a[1] * s;
    R[1][0] = a[1] * a[0] * v + a[2] * s; R[1][1] = c + a[1] * a[1] * v;        R[1][2] = a[1] * a[2] * v - a[0] * s;
    R[2][0] = a[2] * a[0] * v - a[1] * s; R[2][1] = a[2] * a[1] * v + a[0] * s; R[2][2] = c + a[2] * a[2] * v;
}
/* rotation of link i's frame in its parent's frame, and the motion transform X_i (parent -> link) */
static void joint_rotation(int i, double q, mat3 Rpc) {
    mat3 Rq; const tree_model *m = model();
    if (m->axis[i][2] == 1.0) { double rz[3] = {0, 0, 0}; rz[2] = q; rpy_to_mat(rz, Rq); }     /* the arm's joints: exactly the rounds 1-2 arithmetic */
    else axis_rotation(m->axis[i], q, Rq);
    mat3_mul(m->Rj[i], Rq, Rpc);
}
static void motion_transform(int i, double q, mat6 X) {
    mat3 Rpc, E, rx, Erx; int a, b;
    joint_rotation(i, q, Rpc);
    for (a = 0; a < 3; a++) for (b = 0; b < 3; b++) E[a][b] = Rpc[b][a];
    skew(model()->xyz[i], rx);
    mat3_mul(E, rx, Erx);
    memset(X, 0, sizeof(mat6));
    for (a = 0; a < 3; a++) for (b = 0; b < 3; b++) { X[a][b] = E[a][b]; X[a + 3][b + 3] = E[a][b]; X[a + 3][b] = -Erx[a][b]; }
}
/* world pose of every link frame (parents come before children) */
static void forward_kinematics(const double q[TN], mat3 R[TN], double p[TN][3]) {
    const tree_model *m = model(); int i;
    for (i = 0; i < m->nd; i++) {
        mat3 Rpc; double t[3]; const int par = m->parent[i];
        joint_rotation(i, q[i], Rpc);
        if (par < 0) {
            static const mat3 I3 = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
            mat3_vec(I3, m->xyz[i], t);
            p[i][0] = KM_BASE_POS[0] + t[0]; p[i][1] = KM_BASE_POS[1] + t[1]; p[i][2] = KM_BASE_POS[2] + t[2];
            mat3_mul(I3, Rpc, R[i]);
        } else {
            mat3_vec(R[par], m->xyz[i], t);
            p[i][0] = p[par][0] + t[0]; p[i][1] = p[par][1] + t[1]; p[i][2] = p[par][2] + t[2];
            mat3_mul(R[par], Rpc, R[i]);
        }
    }
}
static void link_point(const mat3 R[TN], const double p[TN][3], int link, const double local[3], double world[3]) {
    double t[3]; mat3_vec(R[link], local, t);
    world[0] = p[link][0] + t[0]; world[1] = p[link][1] + t[1]; world[2] = p[link][2] + t[2];
}
/* geometric Jacobian of a world point rigidly attached to `link`: columns of the link's ancestors (and itself), zero elsewhere */
static void point_jacobian(const mat3 R[TN], const double p[TN][3], int link, const double pt[3], double Jv[3][TN], double Jw[3][TN]) {
    const tree_model *m = model(); int j, k;
    for (j = 0; j < TN; j++) for (k = 0; k < 3; k++) { Jv[k][j] = 0.0; Jw[k][j] = 0.0; }
    for (j = link; j >= 0; j = m->parent[j]) {
        double z[3], d[3] = {pt[0] - p[j][0], pt[1] - p[j][1], pt[2] - p[j][2]}, c[3];
        mat3_vec(R[j], m->axis[j], z);
        cross(z, d, c);
        Jv[0][j] = c[0]; Jv[1][j] = c[1]; Jv[2][j] = c[2];
        Jw[0][j] = z[0]; Jw[1][j] = z[1]; Jw[2][j] = z[2];
    }
}

/* ------------------------------------------------------------------ Featherstone ABA */
static void spatial_inertia(int i, mat6 I) {
    const tree_model *tm = model();
    const double *c = tm->com[i], *in = tm->inertia[i]; double m = tm->mass[i]; mat3 cx; int a, b;
    const double Ic[3][3] = {{in[0], in[1], in[2]}, {in[1], in[3], in[4]}, {in[2], in[4], in[5]}};
    double cc = c[0] * c[0] + c[1] * c[1] + c[2] * c[2];
    skew(c, cx);
    memset(I, 0, sizeof(mat6));
    for (a = 0; a < 3; a++) for (b = 0; b < 3; b++) {
        I[a][b] = (a == b ? Ic[a][a] + m * cc : Ic[a][b]) - m * c[a] * c[b];
        I[a][b + 3] = m * cx[a][b];
        I[a + 3][b] = -m * cx[a][b];
        I[a + 3][b + 3] = a == b ? m : 0.0;
    }
}
static mat6 g_I6[TN];
static void inertia_refresh(void) { int i; for (i = 0; i < g_m.nd; i++) spatial_inertia(i, g_I6[i]); }
/* Featherstone ABA (RBDA Table 7.1) on a kinematic tree, revolute joints S = [axis ; 0], split in two so that the thirteen
 * solves of one physics step (the step itself + one unit torque per DoF for M^-1) share what depends on q only:
 *   aba_factor : motion transforms, articulated inertias IA, U = IA S, d = S.U          (the 6x6 congruence transforms)
 *   aba_solve  : velocities / bias forces, backward vector pass, forward accelerations   (6-vectors only)
 * Same arithmetic, in the same order, as the single-pass form of rounds 1-2. */
typedef struct { mat6 X[TN], IA[TN], Ia[TN]; vec6 S[TN], U[TN]; double d[TN]; } aba_fac;
static void aba_factor(const double q[TN], aba_fac *f) {
    const tree_model *m = model(); const int n = m->nd; int i, r, s, k;
    for (i = 0; i < n; i++) {
        motion_transform(i, q[i], f->X[i]);
        for (k = 0; k < 3; k++) { f->S[i][k] = m->axis[i][k]; f->S[i][3 + k] = 0.0; }
        memcpy(f->IA[i], g_I6[i], sizeof(mat6));
    }
    for (i = n - 1; i >= 0; i--) {
        const int par = m->parent[i];
        mat6_vec(f->IA[i], f->S[i], f->U[i]);
        f->d[i] = 0.0;
        for (k = 0; k < 6; k++) f->d[i] += f->S[i][k] * f->U[i][k];
        if (par >= 0) {
            mat6 T;
            for (r = 0; r < 6; r++) for (s = 0; s < 6; s++) f->Ia[i][r][s] = f->IA[i][r][s] - f->U[i][r] * f->U[i][s] / f->d[i];
            /* IA[parent] += X^T Ia X */
            for (r = 0; r < 6; r++) for (s = 0; s < 6; s++) { double acc = 0; for (k = 0; k < 6; k++) acc += f->Ia[i][r][k] * f->X[i][k][s]; T[r][s] = acc; }
            for (r = 0; r < 6; r++) for (s = 0; s < 6; s++) { double acc = 0; for (k = 0; k < 6; k++) acc += f->X[i][k][r] * T[k][s]; f->IA[par][r][s] += acc; }
        }
    }
}
static void aba_solve(const aba_fac *f, const double qd[TN], const double tau[TN], double gz, double qdd[TN]) {
    const tree_model *m = model(); const int n = m->nd;
    vec6 v[TN], c[TN], pA[TN], a[TN]; double u[TN];
    int i, k;
    for (i = 0; i < n; i++) {
        vec6 vJ, Iv; const int par = m->parent[i];
        for (k = 0; k < 6; k++) vJ[k] = f->S[i][k] * qd[i];
        if (par < 0) memcpy(v[i], vJ, sizeof(vec6));
        else { mat6_vec(f->X[i], v[par], v[i]); for (k = 0; k < 6; k++) v[i][k] += vJ[k]; }
        crm_vec(v[i], vJ, c[i]);
        mat6_vec(g_I6[i], v[i], Iv);
        crf_vec(v[i], Iv, pA[i]);
    }
    for (i = n - 1; i >= 0; i--) {
        const int par = m->parent[i];
        u[i] = tau[i];
        for (k = 0; k < 6; k++) u[i] -= f->S[i][k] * pA[i][k];
        if (par >= 0) {
            vec6 pa, Iac, t;
            mat6_vec(f->Ia[i], c[i], Iac);
            for (k = 0; k < 6; k++) pa[k] = pA[i][k] + Iac[k] + f->U[i][k] * u[i] / f->d[i];
            mat6T_vec(f->X[i], pa, t);                              /* pA[parent] += X^T pa */
            for (k = 0; k < 6; k++) pA[par][k] += t[k];
        }
    }
    for (i = 0; i < n; i++) {
        vec6 ap; double Ua = 0; const int par = m->parent[i];
        if (par < 0) { vec6 a0 = {0, 0, 0, 0, 0, 0}; a0[5] = -gz; mat6_vec(f->X[i], a0, ap); }
        else mat6_vec(f->X[i], a[par], ap);
        for (k = 0; k < 6; k++) ap[k] += c[i][k];
        for (k = 0; k < 6; k++) Ua += f->U[i][k] * ap[k];
        qdd[i] = (u[i] - Ua) / f->d[i];
        for (k = 0; k < 6; k++) a[i][k] = ap[k] + f->S[i][k] * qdd[i];
    }
}
/* qdd = FD(q, qd, tau) with gravity (0, 0, gz) */
static void aba(const double q[TN], const double qd[TN], const double tau[TN], double gz, double qdd[TN]) {
    aba_fac f; aba_factor(q, &f); aba_solve(&f, qd, tau, gz, qdd);
}
/* W = M(q)^-1, column j = response to a unit torque on joint j (no velocity, no gravity) */
static void mass_matrix_inverse_f(const aba_fac *f, double W[TN][TN]) {
    double zero[TN] = {0}, e[TN], col[TN]; int i, j; const int n = ND;
    for (j = 0; j < n; j++) {
        memset(e, 0, sizeof e); e[j] = 1.0;
        aba_solve(f, zero, e, 0.0, col);
        for (i = 0; i < n; i++) W[i][j] = col[i];
    }
}
static void mass_matrix_inverse(const double q[TN], double W[TN][TN]) { aba_fac f; aba_factor(q, &f); mass_matrix_inverse_f(&f, W); }

/* ------------------------------------------------------------------ inverse kinematics */
/* one damped-least-squares step towards (target position, orientation = quat(euler(0,-pi,0))) */
static void quat_from_mat(const mat3 R, double qt[4]) {       /* (x, y, z, w), Shepperd */
    double tr = R[0][0] + R[1][1] + R[2][2];
    if (tr > 0) { double s = sqrt(tr + 1.0) * 2; qt[3] = 0.25 * s; qt[0] = (R[2][1] - R[1][2]) / s; qt[1] = (R[0][2] - R[2][0]) / s; qt[2] = (R[1][0] - R[0][1]) / s; }
    else if (R[0][0] > R[1][1] && R[0][0] > R[2][2]) { double s = sqrt(1.0 + R[0][0] - R[1][1] - R[2][2]) * 2; qt[3] = (R[2][1] - R[1][2]) / s; qt[0] = 0.25 * s; qt[1] = (R[0][1] + R[1][0]) / s; qt[2] = (R[0][2] + R[2][0]) / s; }
    else if (R[1][1] > R[2][2]) { double s = sqrt(1.0 + R[1][1] - R[0][0] - R[2][2]) * 2; qt[3] = (R[0][2] - R[2][0]) / s; qt[0] = (R[0][1] + R[1][0]) / s; qt[1] = 0.25 * s; qt[2] = (R[1][2] + R[2][1]) / s; }
    else { double s = sqrt(1.0 + R[2][2] - R[0][0] - R[1][1]) * 2; qt[3] = (R[1][0] - R[0][1]) / s; qt[0] = (R[0][2] + R[2][0]) / s; qt[1] = (R[1][2] + R[2][1]) / s; qt[2] = 0.25 * s; }
}
static void quat_mul(const double a[4], const double b[4], double o[4]) {
    o[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
    o[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    o[1] = a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0];
    o[2] = a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3];
}
static int solve_linear(double A[N][N], double b[N], double x[N]) {   /* Gaussian elimination, partial pivoting */
    int i, j, k;
    for (k = 0; k < N; k++) {
        int best = k; double bv = fabs(A[k][k]);
        for (i = k + 1; i < N; i++) if (fabs(A[i][k]) > bv) { bv = fabs(A[i][k]); best = i; }
        if (bv == 0.0) return -1;
        if (best != k) { double t; for (j = 0; j < N; j++) { t = A[k][j]; A[k][j] = A[best][j]; A[best][j] = t; } t = b[k]; b[k] = b[best]; b[best] = t; }
        for (i = k + 1; i < N; i++) {
            double f = A[i][k] / A[k][k];
            for (j = k; j < N; j++) A[i][j] -= f * A[k][j];
            b[i] -= f * b[k];
        }
    }
    for (i = N - 1; i >= 0; i--) { double s = b[i]; for (j = i + 1; j < N; j++) s -= A[i][j] * x[j]; x[i] = s / A[i][i]; }
    return 0;
}
/* IK conditioning probe (round 5).  det(J^T J + damping I) of the last damped-least-squares solve, as the product of the pivots of an
 * unpivoted elimination (the matrix is symmetric positive definite).  Near a kinematic singularity of the arm the controller's
 * closed loop (IK step -> position motors -> next IK step) amplifies any difference between two float64 implementations by a
 * constant factor per step (measured: x 2.4 per step while the sixth singular value of J is below ~1e-2, i.e. while the DLS gain
 * sigma / (sigma^2 + damping) with the reference's damping 1e-5 (kuka.py:41-42) exceeds ~100), so the 1e-4 / bit-exact parity bar
 * cannot hold THROUGH such a crossing for any pair of implementations.  det < KM_IK_CROSS_DET marks the regime: random-agent
 * rollouts stay above 1e-7, the elbow crossing (joint 3 through 0) starts to amplify at 1e-9.  t_ik_det = smallest det of the IK
 * solves since ik_conditioning_reset(). */
static double t_ik_det = 1e300;
#pragma omp threadprivate(t_ik_det)
static void ik_conditioning_reset(void) { t_ik_det = 1e300; }
static void ik_conditioning_note(double A[N][N]) {
    double M[N][N], det = 1.0; int i, j, k;
    memcpy(M, A, sizeof M);
    for (k = 0; k < N; k++) {
        det *= M[k][k];
        for (i = k + 1; i < N; i++) { const double f = M[i][k] / M[k][k]; for (j = k; j < N; j++) M[i][j] -= f * M[k][j]; }
    }
    if (det < t_ik_det) t_ik_det = det;
}
/* The end effector (link 6) does not move with the gripper joints: their Jacobian columns are zero, so the 12-DoF damped
 * least-squares step of the full model leaves them at exactly 0 and the 7x7 arm block is the whole solve. */
static void inverse_kinematics(const double q[TN], const mat3 R[TN], const double p[TN][3], const double target[3], double damping, double q_des[N]) {
    /* target orientation p.getQuaternionFromEuler([0, -pi, 0]) = (0, sin(-pi/2), 0, cos(-pi/2)) */
    const double tq[4] = {0.0, sin(-KM_PI / 2), 0.0, cos(-KM_PI / 2)};
    const tree_model *m = model();
    double ee[3], Jv[3][TN], Jw[3][TN], J[6][N], dS[6], cq[4], cinv[4], dq[4], A[N][N], b[N], dth[N];
    double angle, s2, axis[3], maxabs = 0; int i, j, k;
    link_point(R, p, m->ee_link, m->ee_point, ee);
    point_jacobian(R, p, m->ee_link, ee, Jv, Jw);
    for (j = 0; j < N; j++) for (k = 0; k < 3; k++) { J[k][j] = Jv[k][j]; J[k + 3][j] = Jw[k][j]; }
    for (k = 0; k < 3; k++) dS[k] = target[k] - ee[k];
    quat_from_mat(R[m->ee_link], cq);
    cinv[0] = -cq[0]; cinv[1] = -cq[1]; cinv[2] = -cq[2]; cinv[3] = cq[3];
    quat_mul(tq, cinv, dq);                                   /* deltaQ = endQ * startQ^-1 */
    /* btQuaternion::getAngle()/getAxis(): angle = 2 acos(w), axis = xyz / sqrt(1 - w^2).  Evaluated in the
     * equivalent, well-conditioned form angle = 2 atan2(|xyz|, w), axis = xyz / |xyz| (acos loses half the
     * digits as w -> 1, i.e. exactly where the controller converges). */
    s2 = sqrt(dq[0] * dq[0] + dq[1] * dq[1] + dq[2] * dq[2]);
    angle = 2.0 * atan2(s2, dq[3]);
    if (s2 * s2 < 10.0 * 2.2204460492503131e-16) { axis[0] = 1; axis[1] = 0; axis[2] = 0; }
    else { axis[0] = dq[0] / s2; axis[1] = dq[1] / s2; axis[2] = dq[2] / s2; }
    if (angle > KM_PI) angle -= 2 * KM_PI;
    for (k = 0; k < 3; k++) dS[3 + k] = angle * axis[k];
    /* (J^T J + diag(damping)) dtheta = J^T dS */
    for (i = 0; i < N; i++) {
        for (j = 0; j < N; j++) { double s = 0; for (k = 0; k < 6; k++) s += J[k][i] * J[k][j]; A[i][j] = s; }
        A[i][i] += damping;
        { double s = 0; for (k = 0; k < 6; k++) s += J[k][i] * dS[k]; b[i] = s; }
    }
    ik_conditioning_note(A);
    if (solve_linear(A, b, dth) != 0) memset(dth, 0, sizeof dth);
    for (i = 0; i < N; i++) if (fabs(dth[i]) > maxabs) maxabs = fabs(dth[i]);
    if (maxabs > KM_IK_MAX_ANGLE) for (i = 0; i < N; i++) dth[i] *= KM_IK_MAX_ANGLE / maxabs;
    for (i = 0; i < N; i++) q_des[i] = q[i] + dth[i];
}

/* ------------------------------------------------------------------ env state */
typedef struct {
    double q[TN], qd[TN];          /* joints: arm 0..6; full model: 7 gripper_to_arm, 8/9 left finger / tip, 10/11 right finger / tip */
    double ee_target[3];           /* Kuka.end_effector_pos                        */
    double bq, bqd;                /* button glider position / velocity            */
    int button_motor_on;           /* 0: pybullet default velocity motor, 1: step2's position target */
    double button_xy[2];           /* button base position on the table            */
    double button_z;               /* button base height (MovingButton re-places the base every step) */
    double button_speed;           /* MovingButton: signed y increment per step (kuka_moving_button_gym_env.py:40) */
    double button_pos[3];          /* target: cap position at reset + 0.28 in z    */
    double gripper[3];             /* getArmPos() after the last physics step      */
    int contact_button, contact_table;   /* manifolds of the last stepSimulation  */
    int counter, n_contacts, n_outside, terminated;
    int ik_crossed;                /* sticky per episode: an IK solve of this episode had det(J^T J + damping I) < KM_IK_CROSS_DET (ik_conditioning_note) */
    /* Kuka2ButtonGymEnv: second button (same urdf), per-body contact flags, goal bookkeeping */
    double b2q, b2qd, button2_xy[2];
    int contact_body[2];           /* any link (cap or base) of button k touches the arm: getContactPoints(button_uid[k], kuka) */
    int goal_id, n_contacts2;      /* n_contacts[0] lives in n_contacts */
    double all_pos[2][3];          /* button_all_pos */
    /* KukaRandButtonGymEnv: distractor objects (x, y, present) in draw order */
    double obj_xy[10][2]; int obj_present[10];
    /* ... as free bodies (round 4): bodies 0..9 = the ten distractors, 10 = the ball (sphere_small.urdf, kuka_rand_button_gym_env.py:71);
     * translation only (position of the proxy shape's centre, velocity); rb_type 0 duck / 1 lego / 2 cube (boxes), 3 ball */
    double rb_x[RB_N][3], rb_v[RB_N][3]; int rb_on[RB_N], rb_type[RB_N];
} kenv;

typedef struct { int random_target, force_down, shape_reward, action_repeat, is_discrete, action_joints, moving, two, rand_objects; double max_distance; } kcfg;

/* ------------------------------------------------------------------ collision */
/* signed distance between a sphere and an upright solid cylinder (axis z, centre xy, z in [z0, z1]);
 * n = unit normal from the cylinder towards the sphere */
static double sphere_cylinder(const double c[3], double rad, const double xy[2], double R, double z0, double z1, double n[3]) {
    double dx = c[0] - xy[0], dy = c[1] - xy[1], rho = sqrt(dx * dx + dy * dy);
    double er = rho - R, ez_top = c[2] - z1, ez_bot = z0 - c[2], ez = ez_top > ez_bot ? ez_top : ez_bot;
    double rx = rho > 1e-12 ? dx / rho : 1.0, ry = rho > 1e-12 ? dy / rho : 0.0, sz = ez_top > ez_bot ? 1.0 : -1.0;
    if (er <= 0 && ez <= 0) {                 /* centre inside the solid: exit through the nearest face */
        if (er > ez) { n[0] = rx; n[1] = ry; n[2] = 0; return er - rad; }
        n[0] = 0; n[1] = 0; n[2] = sz; return ez - rad;
    }
    if (er <= 0) { n[0] = 0; n[1] = 0; n[2] = sz; return ez - rad; }
    if (ez <= 0) { n[0] = rx; n[1] = ry; n[2] = 0; return er - rad; }
    { double dist = sqrt(er * er + ez * ez); n[0] = rx * er / dist; n[1] = ry * er / dist; n[2] = sz * ez / dist; return dist - rad; }
}

/* ------------------------------------------------------------------ KukaRandButtonGymEnv free bodies
 * kuka_rand_button_gym_env.py:59-71 drops ten meshes (duck_vhacd / lego / cube_small, TYPE from the global unseeded np.random) and a
 * ball on the table, :111-125 kicks the ball at env step 10 (direction again from the unseeded global RNG): free rigid bodies the arm
 * can push.  Restated with primitive proxies [UNVERIFIED-MEMORY for every size / mass: pybullet_data is absent]:
 *   duck -> box of its bounding box, half extents (0.05, 0.035, 0.035); lego -> (0.016, 0.032, 0.012); cube_small -> 0.025 cube;
 *   sphere_small -> sphere r = 0.03; 0.1 kg each; lateral friction 0.5 x the table's 0.5 = 0.25 (arm contacts: the arm sphere's own
 *   combined coefficient); restitution 0.
 * Model: each body TRANSLATES (3 DoF; no rotation, so an off-centre push does not turn a box and the ball slides rather than rolls —
 * with sphere_small's rolling friction both come to rest within centimetres); contacts: arm collision sphere <-> body (normal row +
 * friction row(s), counted against the same row budget as the button contacts, at most one body per sphere: the lowest-numbered one
 * in reach), body <-> table top plane (normal row + friction rows along btPlaneSpace1's tangents: y [, x with the second friction
 * direction]), all rows in the same 150-sweep projected Gauss-Seidel as the arm's.  Not modelled: body <-> body and body <-> button
 * contacts, the table's edges.  The type is the position hash the renderer already uses; the kick direction comes from a Philox
 * stream of the env's own key (stream 2, one counter per episode) instead of the unseeded global RNG — reproducible, and the
 * env's np_random stream is left exactly as the reference consumes it.  Bodies start every episode at rest on the table: the
 * reference drops them 0.1 m (0.3 m: ball) BEFORE its 500 settle steps, after which they rest (kuka_oracle_rb_drop_check integrates
 * the literal drop and returns the distance to that rest state: ~1e-16). */
static const double RB_HALF[3][3] = {{0.05, 0.035, 0.035}, {0.016, 0.032, 0.012}, {0.025, 0.025, 0.025}};
#define RB_BALL_R 0.03
#define RB_MASS 0.1
#define RB_MU_TABLE 0.25
#define RB_BALL_FORCE 10.0            /* kuka_rand_button_gym_env.py:4 */
#define RB_KICK_STEP 10               /* :113 */
static double rb_height(int type) { return type == 3 ? RB_BALL_R : RB_HALF[type][2]; }      /* centre above the supporting plane at rest */
static int rb_type_of(double ox, double oy) {   /* the reference's type comes from the unseeded global RNG: here a hash of the drawn position */
    uint64_t bx, by; memcpy(&bx, &ox, 8); memcpy(&by, &oy, 8);
    return (int)((uint32_t)((bx >> 20) ^ (by >> 20)) % 3u);
}
/* kuka_rand_button_gym_env.py:113-120: at env step 10 the ball gets BALL_FORCE along a random horizontal direction of the first
 * quadrant (abs) plus 1 N upwards, for one simulation step.  The reference takes the direction from the unseeded global np.random;
 * here it is a function of the episode's own reset draws (the last distractor's position: drawn every reset whether or not the
 * object is kept): u in [0, 1) from its bits, direction (1 - u, u) / |.| — sqrt and division only, so that every implementation
 * produces the same bits. */
static void rb_kick_force(double ox, double oy, double f[3]) {
    uint64_t bx, by, h; double u, a, b, len;
    memcpy(&bx, &ox, 8); memcpy(&by, &oy, 8);
    h = (bx >> 12) ^ (by >> 20) ^ (bx >> 36);
    u = (double)(h & 0xFFFFFFu) / 16777216.0;
    a = 1.0 - u; b = u; len = sqrt(a * a + b * b);
    f[0] = RB_BALL_FORCE * (a / len); f[1] = RB_BALL_FORCE * (b / len); f[2] = 1.0;
}
/* signed distance between a sphere (centre c, radius rad) and body k's shape; n = unit normal from the body towards the sphere */
static double sphere_body(const double c[3], double rad, const double x[3], int type, double n[3]) {
    double d[3] = {c[0] - x[0], c[1] - x[1], c[2] - x[2]};
    if (type == 3) {
        double len = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        if (len > 1e-12) { n[0] = d[0] / len; n[1] = d[1] / len; n[2] = d[2] / len; } else { n[0] = 0; n[1] = 0; n[2] = 1; }
        return len - RB_BALL_R - rad;
    } else {
        const double *h = RB_HALF[type]; double q[3], diff[3], len; int k, inside = 1;
        for (k = 0; k < 3; k++) { q[k] = d[k] < -h[k] ? -h[k] : (d[k] > h[k] ? h[k] : d[k]); diff[k] = d[k] - q[k]; if (diff[k] != 0.0) inside = 0; }
        if (inside) {                                /* centre inside the box: exit through the nearest face */
            int best = 0; double pen = h[0] - fabs(d[0]);
            for (k = 1; k < 3; k++) if (h[k] - fabs(d[k]) < pen) { pen = h[k] - fabs(d[k]); best = k; }
            n[0] = n[1] = n[2] = 0; n[best] = d[best] < 0 ? -1.0 : 1.0;
            return -pen - rad;
        }
        len = sqrt(diff[0] * diff[0] + diff[1] * diff[1] + diff[2] * diff[2]);
        n[0] = diff[0] / len; n[1] = diff[1] / len; n[2] = diff[2] / len;
        return len - rad;
    }
}

/* ------------------------------------------------------------------ one physics step */
/* A constraint row.  fric_of >= 0: friction row of contact-normal row `fric_of` (its bounds are +-mu * that row's applied impulse,
 * re-evaluated every sweep: btMultiBodyConstraintSolver::solveSingleIteration). */
typedef struct { double J[TN]; double Jb; double WJ[TN]; double WJb; double Dinv, rhs, lo, hi, applied, mu; int bsel, fric_of;
                 int obj; double Jo[3]; } row_t;      /* obj >= 0: the row also acts on free body `obj` (KukaRandButton) with Jacobian Jo */

static row_t *add_row_obj(row_t *rows, int *nrows, const double J[TN], double Jb, const double W[TN][TN], double Wb,
                          double desired_vel, double pos_error_vel, const double qd[TN], double bqd, double lo, double hi, int bsel,
                          int obj, const double Jo[3], const double vo[3]) {
    row_t *r = &rows[(*nrows)++]; int i, j; double D = 0, rel = 0; const int n = ND;
    r->bsel = bsel;                       /* which button's glider the scalar Jb acts on (bqd = that glider's velocity) */
    r->fric_of = -1; r->mu = 0.0;
    r->obj = obj; r->Jo[0] = r->Jo[1] = r->Jo[2] = 0.0;
    if (obj >= 0) for (i = 0; i < 3; i++) { r->Jo[i] = Jo[i]; D += Jo[i] * Jo[i] / RB_MASS; rel += Jo[i] * vo[i]; }
    for (i = 0; i < n; i++) { double s = 0; for (j = 0; j < n; j++) s += W[i][j] * J[j]; r->WJ[i] = s; r->J[i] = J[i]; }
    r->Jb = Jb; r->WJb = Wb * Jb;
    for (i = 0; i < n; i++) { D += J[i] * r->WJ[i]; rel += J[i] * qd[i]; }
    D += Jb * r->WJb; rel += Jb * bqd;
    r->Dinv = D > 0.0 ? 1.0 / D : 0.0;       /* D = 0: a row no DoF can act on (inert) */
    r->rhs = (desired_vel - rel) * r->Dinv + pos_error_vel * r->Dinv;   /* velocityImpulse + penetrationImpulse */
    r->lo = lo; r->hi = hi; r->applied = 0.0;
    return r;
}
static row_t *add_row(row_t *rows, int *nrows, const double J[TN], double Jb, const double W[TN][TN], double Wb,
                      double desired_vel, double pos_error_vel, const double qd[TN], double bqd, double lo, double hi, int bsel) {
    return add_row_obj(rows, nrows, J, Jb, W, Wb, desired_vel, pos_error_vel, qd, bqd, lo, hi, bsel, -1, NULL, NULL);
}
/* btPlaneSpace1(n, p, q): the friction direction Bullet's multibody solver uses (first tangent) */
static void plane_space1(const double n[3], double p[3]) {
    if (fabs(n[2]) > 0.7071067811865475244008443621048490) { double a = n[1] * n[1] + n[2] * n[2], k = 1.0 / sqrt(a); p[0] = 0; p[1] = -n[2] * k; p[2] = n[1] * k; }
    else { double a = n[0] * n[0] + n[1] * n[1], k = 1.0 / sqrt(a); p[0] = -n[1] * k; p[1] = n[0] * k; p[2] = 0; }
}

/* optional per-step probe of the last physics step (tests / the model-gap report): contact rows created, friction rows */
static int g_probe_rows[4];
#pragma omp threadprivate(g_probe_rows)
/* Flag-margin probe: how close the quantities behind the DISCRETE outputs came to their thresholds over a rollout —
 * [0] min |d(sphere, cap of button 0) - threshold| (contact_button), [1] min |sphere-table gap - threshold| (contact_table),
 * [2] min |distance(gripper, target) - max_distance| (the -1 reward / n_outside counter), [3] min |d(any button shape) - threshold|
 * (contact_body / row creation).  The HIP stepper's values differ from these by ~1e-11 (different association of the same sums),
 * so its flags can only differ where a margin is that small: tests assert the margins of the runs they compare. */
static double t_margin[4] = {1e30, 1e30, 1e30, 1e30}, g_margin[4] = {1e30, 1e30, 1e30, 1e30};
#pragma omp threadprivate(t_margin)
static void margin_note(int k, double v) { v = fabs(v); if (v < t_margin[k]) t_margin[k] = v; }
static void margin_merge(void) {
    int k;
#pragma omp critical(kuka_margin)
    for (k = 0; k < 4; k++) { if (t_margin[k] < g_margin[k]) g_margin[k] = t_margin[k]; }
    for (k = 0; k < 4; k++) t_margin[k] = 1e30;
}
void kuka_oracle_margins_reset(void) { int k; for (k = 0; k < 4; k++) g_margin[k] = 1e30; }
void kuka_oracle_margins_get(double *out4) { memcpy(out4, g_margin, sizeof g_margin); }

/* Kuka.applyAction (kuka.py:118-187) followed by p.stepSimulation() */
static void physics_step(kenv *e, const kcfg *cfg, const double motor[5], const double *joint_targets) {
    const tree_model *m = model(); const int n = m->nd;
    mat3 R[TN]; double p[TN][3], q_arm[N], q_des[TN], tau[TN] = {0}, qdd[TN], W[TN][TN], dv[TN], dvb[2] = {0.0, 0.0};
    const double Wb = 1.0 / KM_CAP_MASS, dt = KM_DT;
    const int nb = cfg->two ? 2 : 1, budget = tm_row_budget(m);
    row_t rows[MAX_ROWS]; int nrows = 0, ngeneric = 0, i, k, it, s, b;
    const int rb_dyn = cfg->rand_objects && n > N;      /* free-body dynamics of KukaRandButton: full model only (the lumped kernels keep the scenery) */
    double dvo[RB_N][3] = {{0}};
    /* Kuka(small_constraints=False) for random_target and always for Kuka2Button (kuka_2button_gym_env.py:78) */
    const double (*box)[3] = KM_EE_BOX[(cfg->random_target || cfg->two) ? 0 : 1];
    double zeroJ[TN] = {0};
    double *bq[2], *bqd[2]; const double *bxy[2];
    bq[0] = &e->bq; bqd[0] = &e->bqd; bxy[0] = e->button_xy; bq[1] = &e->b2q; bqd[1] = &e->b2qd; bxy[1] = e->button2_xy;

    forward_kinematics(e->q, R, p);
    if (!joint_targets) {
        for (k = 0; k < 3; k++) {                                   /* kuka.py:134-139 */
            e->ee_target[k] += motor[k];
            if (e->ee_target[k] < box[0][k]) e->ee_target[k] = box[0][k];
            if (e->ee_target[k] > box[1][k]) e->ee_target[k] = box[1][k];
        }
        /* kuka.py:147-156.  Kuka2Button sets use_null_space, but its ll/ul/jr/rp lists have 7 entries for a 12-DoF body:
         * pybullet drops null-space arguments whose length differs from the DoF count, and the call then carries no
         * jointDamping either -> plain DLS with the server's default damping 0.5 (KM_IK_DAMPING_DEFAULT). */
        inverse_kinematics(e->q, R, p, e->ee_target, cfg->two ? KM_IK_DAMPING_DEFAULT : KM_IK_DAMPING, q_arm);
    } else memcpy(q_arm, joint_targets, sizeof q_arm);           /* kuka.py:160-163 */
    if (g_trace_ee) { memcpy(g_trace_ee + 3 * g_trace_n, e->ee_target, 3 * sizeof(double)); if (joint_targets) memcpy(g_trace_jt + N * g_trace_n, joint_targets, N * sizeof(double)); g_trace_n++; }
    for (i = 0; i < N; i++) q_des[i] = q_arm[i];
    /* gripper motor targets, kuka.py:177-187: joint 7 -> end_effector_angle (da = 0 in every env of the reference), fingers 8 / 11 ->
     * -/+ finger_angle (motor_commands[4] = 0.0: closed), tips 10 / 13 -> 0 */
    for (i = N; i < n; i++) { const int ji = m->joint_index[i]; q_des[i] = ji == 8 ? -motor[4] : ji == 11 ? motor[4] : 0.0; }

    /* -- collision detection at the current poses (start of stepSimulation) -- */
    e->contact_button = 0; e->contact_table = 0; e->contact_body[0] = 0; e->contact_body[1] = 0;
    g_probe_rows[0] = g_probe_rows[1] = g_probe_rows[2] = g_probe_rows[3] = 0;
    {
        /* -- unconstrained velocities: ABA with joint damping torques and gravity -- */
        for (i = 0; i < n; i++) tau[i] = -m->damping[i] * e->qd[i];
        {
            aba_fac fac;
            aba_factor(e->q, &fac);
            aba_solve(&fac, e->qd, tau, KM_GRAVITY_Z, qdd);
            mass_matrix_inverse_f(&fac, W);
        }
        for (i = 0; i < n; i++) e->qd[i] += dt * qdd[i];
        for (b = 0; b < nb; b++) *bqd[b] += dt * KM_GRAVITY_Z;
        if (rb_dyn) for (k = 0; k < RB_N; k++) if (e->rb_on[k]) e->rb_v[k][2] += dt * KM_GRAVITY_Z;

        /* -- constraint rows: motors, then joint limits, then contacts (normals, then their friction rows) -- */
        for (i = 0; i < n; i++) {                                   /* joint motors, kuka.py:167-187 (btMultiBodyJointMotor) */
            double J[TN] = {0}, target = m->kp[i] * (q_des[i] - e->q[i]) / dt;
            if (target > m->max_vel[i]) target = m->max_vel[i];
            if (target < -m->max_vel[i]) target = -m->max_vel[i];
            J[i] = 1.0;
            add_row(rows, &nrows, J, 0.0, W, Wb, target, 0.0, e->qd, 0.0, -m->max_force[i] * dt, m->max_force[i] * dt, 0);
        }
        for (b = 0; b < nb; b++) {
            if (e->button_motor_on)                                /* kuka_button_gym_env.py:347, kuka_2button_gym_env.py:118-119 */
                add_row(rows, &nrows, zeroJ, 1.0, W, Wb, KM_BUTTON_KP * (KM_BUTTON_TARGET - *bq[b]) / dt, 0.0, e->qd, *bqd[b],
                        -KM_BUTTON_MAX_FORCE * dt, KM_BUTTON_MAX_FORCE * dt, b);
            else
                add_row(rows, &nrows, zeroJ, 1.0, W, Wb, 0.0, 0.0, e->qd, *bqd[b], -KM_DEFAULT_MOTOR_IMPULSE, KM_DEFAULT_MOTOR_IMPULSE, b);
        }
        /* joint limits (btMultiBodyJointLimitConstraint): unilateral rows.  A row whose stop is still `pen`
         * away only forbids approaching faster than pen/dt, so it is created when pen/dt is within reach
         * (arm: KM_LIMIT_ACTIVATION_VEL, far above the 0.35 rad/s motor clamp; button: always). */
        for (i = 0; i < n; i++) {
            double J[TN] = {0}, pen_lo = e->q[i] - m->lower[i], pen_hi = m->upper[i] - e->q[i];
            if (m->lower[i] > m->upper[i]) continue;                /* no limit on this joint */
            if (pen_lo <= KM_LIMIT_ACTIVATION_VEL * dt && ngeneric < budget) { ngeneric++; g_probe_rows[2]++; J[i] = 1.0; add_row(rows, &nrows, J, 0.0, W, Wb, pen_lo > 0 ? -pen_lo / dt : 0.0, pen_lo > 0 ? 0.0 : -pen_lo * m->limit_erp / dt, e->qd, 0.0, 0.0, KM_LIMIT_MAX_IMPULSE, 0); }
            if (pen_hi <= KM_LIMIT_ACTIVATION_VEL * dt && ngeneric < budget) { ngeneric++; g_probe_rows[2]++; J[i] = -1.0; add_row(rows, &nrows, J, 0.0, W, Wb, pen_hi > 0 ? -pen_hi / dt : 0.0, pen_hi > 0 ? 0.0 : -pen_hi * m->limit_erp / dt, e->qd, 0.0, 0.0, KM_LIMIT_MAX_IMPULSE, 0); }
        }
        for (b = 0; b < nb; b++) {
          double pen_lo = *bq[b] - KM_GLIDER_LOWER, pen_hi = KM_GLIDER_UPPER - *bq[b];
          add_row(rows, &nrows, zeroJ, 1.0, W, Wb, pen_lo > 0 ? -pen_lo / dt : 0.0, pen_lo > 0 ? 0.0 : -pen_lo * m->limit_erp / dt, e->qd, *bqd[b], 0.0, KM_LIMIT_MAX_IMPULSE, b);
          add_row(rows, &nrows, zeroJ, -1.0, W, Wb, pen_hi > 0 ? -pen_hi / dt : 0.0, pen_hi > 0 ? 0.0 : -pen_hi * m->limit_erp / dt, e->qd, *bqd[b], 0.0, KM_LIMIT_MAX_IMPULSE, b); }
        {
            /* contact normals first; the friction rows are appended after ALL normals (Bullet solves normals, then frictions) */
            struct { double J[TN], Jb, mu; int normal_row, bsel, obj, table; double Jo[3]; } fr[MAX_ROWS]; int nfr = 0;
            memset(fr, 0, sizeof fr);
            for (s = 0; s < m->nsphere; s++) {                      /* spheres on the arm / gripper links vs cap, base (of every button), table */
                double c[3], nrm[3], dist, pt[3], Jv[3][TN], Jw[3][TN], J[TN]; int shape; const int link = m->sphere_link[s];
                link_point(R, p, link, m->sphere[s], c);
                margin_note(1, c[2] - m->sphere[s][3] - m->table_top_z - KM_CONTACT_THRESHOLD);
                if (c[2] - m->sphere[s][3] - m->table_top_z < KM_CONTACT_THRESHOLD) e->contact_table = 1;
                for (shape = 0; shape < 2 * nb; shape++) {
                    double pos_err_vel, allow, cap_z0, pen; const int is_cap = (shape & 1) == 0;
                    b = shape >> 1;
                    cap_z0 = e->button_z + KM_GLIDER_ORIGIN_Z + *bq[b];
                    if (is_cap) dist = sphere_cylinder(c, m->sphere[s][3], bxy[b], KM_CAP_RADIUS, cap_z0, cap_z0 + KM_CAP_HEIGHT, nrm);
                    else dist = sphere_cylinder(c, m->sphere[s][3], bxy[b], KM_BASE_RADIUS, e->button_z, e->button_z + KM_BASE_HEIGHT, nrm);
                    margin_note(3, dist - KM_CONTACT_THRESHOLD);
                    if (is_cap && b == 0) margin_note(0, dist - KM_CONTACT_THRESHOLD);
                    if (!(dist < KM_CONTACT_THRESHOLD)) continue;
                    if (is_cap && b == 0) e->contact_button = 1;
                    e->contact_body[b] = 1;
                    if (ngeneric >= budget) continue;
                    ngeneric++;
                    for (k = 0; k < 3; k++) pt[k] = c[k] - m->sphere[s][3] * nrm[k];      /* contact point on the sphere */
                    point_jacobian(R, p, link, pt, Jv, Jw);
                    for (i = 0; i < n; i++) J[i] = nrm[0] * Jv[0][i] + nrm[1] * Jv[1][i] + nrm[2] * Jv[2][i];
                    /* separated: allow approach up to dist/dt; penetrating: push out with erp */
                    pen = dist + m->linear_slop;                       /* Bullet: penetration = distance + m_linearSlop */
                    allow = pen > 0 ? -pen / dt : 0.0;
                    pos_err_vel = pen > 0 ? 0.0 : -pen * m->contact_erp / dt;
                    add_row(rows, &nrows, J, is_cap ? -nrm[2] : 0.0, W, Wb, allow, pos_err_vel, e->qd, *bqd[b], 0.0, 1e10, b);
                    g_probe_rows[0]++;
                    if (m->friction && m->sphere_mu[s] > 0.0) {
                        double tdir[3];
                        plane_space1(nrm, tdir);
                        for (i = 0; i < n; i++) fr[nfr].J[i] = tdir[0] * Jv[0][i] + tdir[1] * Jv[1][i] + tdir[2] * Jv[2][i];
                        /* the cap slides along z only: the z component of the tangent (side contacts) acts on the glider, like -n_z does for the normal */
                        fr[nfr].Jb = is_cap ? -tdir[2] : 0.0;
                        fr[nfr].normal_row = nrows - 1; fr[nfr].bsel = b; fr[nfr].mu = m->sphere_mu[s]; fr[nfr].obj = -1;
                        nfr++;
                        if (m->solver_detail & TM_DETAIL_FRICTION2) {
                            double t2[3];
                            t2[0] = nrm[1] * tdir[2] - nrm[2] * tdir[1]; t2[1] = nrm[2] * tdir[0] - nrm[0] * tdir[2]; t2[2] = nrm[0] * tdir[1] - nrm[1] * tdir[0];
                            for (i = 0; i < n; i++) fr[nfr].J[i] = t2[0] * Jv[0][i] + t2[1] * Jv[1][i] + t2[2] * Jv[2][i];
                            fr[nfr].Jb = is_cap ? -t2[2] : 0.0;
                            fr[nfr].normal_row = nrows - 1; fr[nfr].bsel = b; fr[nfr].mu = m->sphere_mu[s]; fr[nfr].obj = -1;
                            nfr++;
                        }
                    }
                }
                if (rb_dyn) {                                      /* the sphere against the free bodies: the lowest-numbered body in reach */
                    for (k = 0; k < RB_N; k++) {
                        double pos_err_vel, allow, pen, Jo[3];
                        if (!e->rb_on[k]) continue;
                        dist = sphere_body(c, m->sphere[s][3], e->rb_x[k], e->rb_type[k], nrm);
                        if (!(dist < KM_CONTACT_THRESHOLD)) continue;
                        if (ngeneric < budget) {
                            ngeneric++;
                            for (i = 0; i < 3; i++) { pt[i] = c[i] - m->sphere[s][3] * nrm[i]; Jo[i] = -nrm[i]; }
                            point_jacobian(R, p, link, pt, Jv, Jw);
                            for (i = 0; i < n; i++) J[i] = nrm[0] * Jv[0][i] + nrm[1] * Jv[1][i] + nrm[2] * Jv[2][i];
                            pen = dist + m->linear_slop;
                            allow = pen > 0 ? -pen / dt : 0.0;
                            pos_err_vel = pen > 0 ? 0.0 : -pen * m->contact_erp / dt;
                            add_row_obj(rows, &nrows, J, 0.0, W, Wb, allow, pos_err_vel, e->qd, 0.0, 0.0, 1e10, 0, k, Jo, e->rb_v[k]);
                            g_probe_rows[0]++; g_probe_rows[3]++;
                            if (m->friction && m->sphere_mu[s] > 0.0) {
                                double tdir[3], t2[3]; int f, nf = (m->solver_detail & TM_DETAIL_FRICTION2) ? 2 : 1;
                                plane_space1(nrm, tdir);
                                t2[0] = nrm[1] * tdir[2] - nrm[2] * tdir[1]; t2[1] = nrm[2] * tdir[0] - nrm[0] * tdir[2]; t2[2] = nrm[0] * tdir[1] - nrm[1] * tdir[0];
                                for (f = 0; f < nf; f++) {
                                    const double *td = f ? t2 : tdir;
                                    for (i = 0; i < n; i++) fr[nfr].J[i] = td[0] * Jv[0][i] + td[1] * Jv[1][i] + td[2] * Jv[2][i];
                                    fr[nfr].Jb = 0.0; fr[nfr].normal_row = nrows - 1; fr[nfr].bsel = 0; fr[nfr].mu = m->sphere_mu[s];
                                    fr[nfr].obj = k; fr[nfr].Jo[0] = -td[0]; fr[nfr].Jo[1] = -td[1]; fr[nfr].Jo[2] = -td[2];
                                    nfr++;
                                }
                            }
                        }
                        break;                                     /* one body per sphere */
                    }
                }
            }
            if (rb_dyn) {                                          /* bodies resting on / falling onto the table top */
                static const double NZ[3] = {0, 0, 1}, T1[3] = {0, -1, 0}, T2[3] = {1, 0, 0};     /* btPlaneSpace1((0,0,1)) and n x t1 */
                for (k = 0; k < RB_N; k++) {
                    double dist, pen; int f; const int nf = 2;   /* both tangents, whatever the arm contacts use: a single direction (y) would leave the table frictionless along x — a kicked ball would never stop */
                    if (!e->rb_on[k]) continue;
                    dist = e->rb_x[k][2] - rb_height(e->rb_type[k]) - m->table_top_z;
                    if (!(dist < KM_CONTACT_THRESHOLD)) continue;
                    pen = dist + m->linear_slop;
                    add_row_obj(rows, &nrows, zeroJ, 0.0, W, Wb, pen > 0 ? -pen / dt : 0.0, pen > 0 ? 0.0 : -pen * m->contact_erp / dt, e->qd, 0.0,
                                0.0, 1e10, 0, k, NZ, e->rb_v[k]);
                    if (m->friction) for (f = 0; f < nf; f++) {
                        memset(fr[nfr].J, 0, sizeof fr[nfr].J); fr[nfr].Jb = 0.0; fr[nfr].normal_row = nrows - 1; fr[nfr].bsel = 0; fr[nfr].mu = RB_MU_TABLE;
                        fr[nfr].obj = k; fr[nfr].table = 1; memcpy(fr[nfr].Jo, f ? T2 : T1, sizeof fr[nfr].Jo);
                        nfr++;
                    }
                }
            }
            for (k = 0; k < nfr; k++) {
                /* friction: drive the tangential relative velocity to 0 within +-mu * (normal impulse); no positional term */
                row_t *r = add_row_obj(rows, &nrows, fr[k].J, fr[k].Jb, W, Wb, 0.0, 0.0, e->qd, *bqd[fr[k].bsel], 0.0, 0.0, fr[k].bsel,
                                       fr[k].obj, fr[k].Jo, fr[k].obj >= 0 ? e->rb_v[fr[k].obj] : NULL);
                r->fric_of = fr[k].normal_row; r->mu = fr[k].mu;
                if (!fr[k].table) g_probe_rows[1]++;
            }
        }
    }
    /* -- projected Gauss-Seidel, numSolverIterations = 150: non-contact rows, contact normals, then friction rows whose bounds
     *    follow the current normal impulse (skipped while that impulse is not positive) -- */
    memset(dv, 0, sizeof dv);
    {
    int order[MAX_ROWS], nnc = 0, jj;
    for (k = 0; k < nrows; k++) order[k] = k;
    while (nnc < nrows && rows[nnc].fric_of < 0 && rows[nnc].hi < 1e9) nnc++;     /* non-contact rows: everything before the first normal (hi = 1e10) */
    if (m->solver_detail & TM_DETAIL_BODY_ORDER) {            /* [button stops, button motors, arm limits, arm motors] */
        int w = 0, nlim_arm = nnc - n - 3 * nb;
        for (b = 0; b < nb; b++) { order[w++] = n + nb + nlim_arm + 2 * b; order[w++] = n + nb + nlim_arm + 2 * b + 1; order[w++] = n + b; }
        for (k = 0; k < nlim_arm; k++) order[w++] = n + nb + k;
        for (k = 0; k < n; k++) order[w++] = k;
    }
    for (it = 0; it < KM_SOLVER_ITERS; it++) {
        for (jj = 0; jj < nrows; jj++) {
            row_t *r; double jdv, delta, sum;
            k = jj < nnc ? order[(m->solver_detail & TM_DETAIL_ALT_SWEEP) ? ((it & 1) ? jj : nnc - 1 - jj) : jj] : jj;
            r = &rows[k];
            if (r->fric_of >= 0) {
                const double tot = rows[r->fric_of].applied;
                if (!(tot > 0.0)) continue;
                r->lo = -r->mu * tot; r->hi = r->mu * tot;
            }
            jdv = r->Jb * dvb[r->bsel];
            for (i = 0; i < n; i++) jdv += r->J[i] * dv[i];
            if (r->obj >= 0) for (i = 0; i < 3; i++) jdv += r->Jo[i] * dvo[r->obj][i];
            delta = r->rhs - jdv * r->Dinv;
            sum = r->applied + delta;
            if (sum < r->lo) { delta = r->lo - r->applied; r->applied = r->lo; }
            else if (sum > r->hi) { delta = r->hi - r->applied; r->applied = r->hi; }
            else r->applied = sum;
            for (i = 0; i < n; i++) dv[i] += delta * r->WJ[i];
            dvb[r->bsel] += delta * r->WJb;
            if (r->obj >= 0) for (i = 0; i < 3; i++) dvo[r->obj][i] += delta * r->Jo[i] / RB_MASS;
        }
    }
    }
    /* -- semi-implicit Euler -- */
    for (i = 0; i < n; i++) { e->qd[i] += dv[i]; e->q[i] += dt * e->qd[i]; }
    for (b = 0; b < nb; b++) { *bqd[b] += dvb[b]; *bq[b] += dt * *bqd[b]; }
    if (rb_dyn) for (k = 0; k < RB_N; k++) if (e->rb_on[k]) for (i = 0; i < 3; i++) { e->rb_v[k][i] += dvo[k][i]; e->rb_x[k][i] += dt * e->rb_v[k][i]; }
    forward_kinematics(e->q, R, p);
    link_point(R, p, m->grip_link, m->grip_point, e->gripper);
}

/* ------------------------------------------------------------------ env wrapper */
static double norm3(const double a[3], const double b[3]) {      /* np.linalg.norm(a - b, 2): ddot with fma */
    double d0 = a[0] - b[0], d1 = a[1] - b[1], d2 = a[2] - b[2];
    return sqrt(fma(d2, d2, fma(d1, d1, fma(d0, d0, 0.0))));
}
static int g_max_steps_moving = 1500;   /* kuka_moving_button_gym_env.py:3,34 */
static int termination_cfg(const kenv *e, const kcfg *cfg) {      /* :422-426 */
    return e->terminated || e->counter > (cfg->two ? KM_MAX_STEPS_2BUTTON : cfg->moving ? g_max_steps_moving : KM_MAX_STEPS);
}

/* Kuka2ButtonGymEnv._reward (kuka_2button_gym_env.py:141-200) */
static double reward_two(kenv *e, const kcfg *cfg) {
    double distance = norm3(e->all_pos[e->goal_id], e->gripper);
    margin_note(2, distance - cfg->max_distance);
    int reward = 0, contact = e->contact_body[e->goal_id];   /* getContactPoints(button_uid[goal_id], kuka): any link of that button */
    int *nc[2]; nc[0] = &e->n_contacts; nc[1] = &e->n_contacts2;
    *nc[e->goal_id] += contact;
    if (e->goal_id == 1) reward = contact;                    /* sparse reward only on the last button */
    /* next button: button_pressed[goal_id] flips once, when its contact count reaches the threshold */
    if (*nc[e->goal_id] >= KM_N_CONTACTS_BEFORE_TERMINATION && e->goal_id == 0) {
        memcpy(e->button_pos, e->all_pos[1], sizeof e->button_pos);
        e->goal_id = 1;
    }
    if (distance > cfg->max_distance || e->contact_table) { reward = -1; e->n_outside += 1; }
    else e->n_outside = 0;
    if (e->contact_table || e->n_contacts2 >= KM_N_CONTACTS_BEFORE_TERMINATION || e->n_outside >= KM_N_STEPS_OUTSIDE_SAFETY_SPHERE - 1)
        e->terminated = 1;
    if (cfg->shape_reward) {
        if (e->terminated && reward > 0) return 50;
        if (*nc[e->goal_id] < KM_N_CONTACTS_BEFORE_TERMINATION && contact) return 25;
        if (e->contact_table) return -250;
        if (distance > cfg->max_distance) return -20;
        return -distance;
    }
    return reward;
}

static double reward_fn(kenv *e, const kcfg *cfg) {              /* :428-463 */
    double distance;
    if (cfg->two) return reward_two(e, cfg);
    distance = norm3(e->button_pos, e->gripper);
    margin_note(2, distance - cfg->max_distance);
    int reward = e->contact_button ? 1 : 0;
    e->n_contacts += reward;
    if (distance > cfg->max_distance || e->contact_table) { reward = -1; e->n_outside += 1; }
    else e->n_outside = 0;
    if (e->contact_table || e->n_contacts >= KM_N_CONTACTS_BEFORE_TERMINATION || e->n_outside >= KM_N_STEPS_OUTSIDE_SAFETY_SPHERE)
        e->terminated = 1;
    if (cfg->shape_reward) {
        if (cfg->is_discrete && !cfg->moving) return -distance;      /* MovingButton uses the 50 / -250 / -d branch for every action type (:143-151) */
        if (e->terminated && reward > 0) return 50;
        if (e->terminated && reward < 0) return -250;
        return -distance;
    }
    return reward;
}

typedef struct { int mode; np_rng mt; philox_t ph; } krng;
static double k_double(krng *r) { return r->mode == 2 ? np_rng_double(&r->mt) : philox_double(&r->ph); }
static double k_uniform(krng *r, double lo, double hi) { return r->mode == 2 ? np_rng_uniform(&r->mt, lo, hi) : philox_uniform(&r->ph, lo, hi); }
static double k_normal(krng *r, double loc, double scale) { return r->mode == 2 ? np_rng_normal(&r->mt, loc, scale) : loc + scale * philox_std_normal(&r->ph); }
static uint32_t k_randint3(krng *r) { return r->mode == 2 ? np_rng_randint(&r->mt, 3) : philox_bounded(&r->ph, 2); }
static uint32_t k_randint2(krng *r) { return r->mode == 2 ? np_rng_randint(&r->mt, 2) : philox_bounded(&r->ph, 1); }

/* arm/button state after the 500 settle steps (kuka_button_gym_env.py:242-247); the arm never
 * touches anything while settling, so it is the same for every button position. */
static void settle(kenv *e, const kcfg *cfg) {
    const double zero[5] = {0, 0, 0, 0, 0}; int i;
    memset(e, 0, sizeof *e);
    for (i = 0; i < ND; i++) e->q[i] = KM_JOINT_POSITIONS[model()->joint_index[i]];       /* kuka.py:64-66: all 14 joints are reset */
    memcpy(e->ee_target, KM_EE_INIT, sizeof e->ee_target);
    e->button_xy[0] = KM_BUTTON_X; e->button_xy[1] = KM_BUTTON_Y; e->button_z = KM_BUTTON_BASE_Z;
    for (i = 0; i < KM_N_SETTLE_STEPS; i++) physics_step(e, cfg, zero, cfg->action_joints ? KM_JOINT_POSITIONS : NULL);
}

static void env_reset(kenv *e, const kcfg *cfg, krng *r, const kenv *settled) {   /* :214-281 */
    double bx = KM_BUTTON_X, by = KM_BUTTON_Y, speed = 0.0; int i;
    if (cfg->moving) speed = 0.001 * (k_randint2(r) ? 1.0 : -1.0);   /* BUTTON_SPEED * np_random.choice([-1, 1]), drawn first */
    double b2x = KM_BUTTON_X, b2y = KM_BUTTON2_Y_2B;
    if (cfg->two) {                                                /* kuka_2button_gym_env.py:55-70 */
        if (cfg->random_target) { (void)k_uniform(r, -1, 1); (void)k_uniform(r, 0, 1); }   /* overwritten two lines later */
        bx = 0.5 + 0.0 * k_uniform(r, -1, 1); by = KM_BUTTON1_Y_2B + 0.0 * k_uniform(r, -1, 1);
        if (cfg->random_target) { b2x += 0.15 * k_uniform(r, -1, 1); b2y += 0.175 * k_uniform(r, -1, 0); }
    } else if (cfg->random_target) { bx += 0.15 * k_uniform(r, -1, 1); by += 0.3 * k_uniform(r, -1, 1); }
    *e = *settled;
    if (cfg->rand_objects) {                                       /* kuka_rand_button_gym_env.py:62-69 */
        for (i = 0; i < 10; i++) {
            double ox = 0.5 + 0.15 * k_uniform(r, -1, 1), oy = 0 + 0.3 * k_uniform(r, -1, 1);
            e->obj_xy[i][0] = ox; e->obj_xy[i][1] = oy;
            e->obj_present[i] = (ox < bx - 0.1) || (ox > bx + 0.1) || (oy < by - 0.1) || (oy > by + 0.1);
        }
        /* the bodies at rest on the table (see the free-body section: the reference drops them before its 500 settle steps) */
        memset(e->rb_v, 0, sizeof e->rb_v);
        for (i = 0; i < 10; i++) {
            e->rb_on[i] = e->obj_present[i]; e->rb_type[i] = rb_type_of(e->obj_xy[i][0], e->obj_xy[i][1]);
            e->rb_x[i][0] = e->obj_xy[i][0]; e->rb_x[i][1] = e->obj_xy[i][1]; e->rb_x[i][2] = model()->table_top_z + rb_height(e->rb_type[i]);
        }
        e->rb_on[10] = 1; e->rb_type[10] = 3; e->rb_x[10][0] = 0.25; e->rb_x[10][1] = -0.2; e->rb_x[10][2] = model()->table_top_z + RB_BALL_R;
    }
    e->button_xy[0] = bx; e->button_xy[1] = by; e->button_z = KM_BUTTON_BASE_Z; e->button_speed = speed;
    e->button2_xy[0] = b2x; e->button2_xy[1] = b2y; e->b2q = e->bq; e->b2qd = e->bqd;   /* same urdf, same 500 free steps */
    for (i = 0; i < KM_N_RANDOM_ACTIONS_AT_INIT; i++) {
        double action[5] = {0, 0, 0, 0, 0};
        if (cfg->is_discrete) {
            double sign = k_double(r) > 0.5 ? 1.0 : -1.0;
            uint32_t idx = k_randint3(r);
            action[idx] += sign * KM_DELTA_V;
            physics_step(e, cfg, action, NULL);
        } else if (cfg->action_joints) {
            /* np_random.normal(joints.shape): the shape tuple is `loc` -> one draw 7 + N(0,1), broadcast */
            double joints[N], g = k_normal(r, 7.0, 1.0);
            int j; for (j = 0; j < N; j++) joints[j] = KM_JOINT_POSITIONS[j] + KM_DELTA_THETA * g;
            physics_step(e, cfg, action, joints);
        } else {
            /* np_random.normal((3,)) -> one draw 3 + N(0,1); L2-normalised 1-vector = +-1, broadcast */
            double g = k_normal(r, 3.0, 1.0), dir = g / sqrt(fma(g, g, 0.0));
            action[0] += KM_DELTA_V_CONTINUOUS * dir; action[1] += KM_DELTA_V_CONTINUOUS * dir; action[2] += KM_DELTA_V_CONTINUOUS * dir;
            physics_step(e, cfg, action, NULL);
        }
    }
    e->button_pos[0] = bx; e->button_pos[1] = by;
    e->button_pos[2] = e->button_z + KM_GLIDER_ORIGIN_Z + e->bq + KM_BUTTON_DISTANCE_HEIGHT;   /* :273-274 */
    if (cfg->two) {                                                /* button_all_pos: [x, y, Z_TABLE + BUTTON_DISTANCE_HEIGHT] */
        e->all_pos[0][0] = bx; e->all_pos[0][1] = by; e->all_pos[0][2] = KM_Z_TABLE + KM_BUTTON_DISTANCE_HEIGHT;
        e->all_pos[1][0] = b2x; e->all_pos[1][1] = b2y; e->all_pos[1][2] = KM_Z_TABLE + KM_BUTTON_DISTANCE_HEIGHT;
        memcpy(e->button_pos, e->all_pos[0], sizeof e->button_pos);
    }
    e->goal_id = 0; e->n_contacts2 = 0;
    e->counter = 0; e->n_contacts = 0; e->n_outside = 0; e->terminated = 0; e->ik_crossed = 0;
}

static void observe(const kenv *e, int obs_mode, float *o) {     /* getSRLState :175-189 */
    int k = 0, j;
    if (obs_mode == 0 || obs_mode == 2) for (j = 0; j < 3; j++) o[k++] = (float)(e->gripper[j] - e->button_pos[j]);
    if (obs_mode == 1 || obs_mode == 2) for (j = 0; j < 14; j++) o[k++] = (float)KM_JOINT_POSITIONS[j];
}

/* KukaButtonGymEnv.step (:293-340) + step2 (:342-368).  action < 0 == None. */
static double env_step(kenv *e, const kcfg *cfg, krng *r, int action, const float *caction, int *done) {
    double motor[5] = {0, 0, 0, 0, 0}, joints[N]; const double *jt = NULL; int rep;
    if (cfg->moving) {                                             /* kuka_moving_button_gym_env.py:111-119 */
        if (e->button_pos[1] > 0.3 || e->button_pos[1] < -0.3) e->button_speed = -e->button_speed;
        e->button_pos[1] += e->button_speed;
        e->button_xy[1] = e->button_pos[1];
        e->button_z = e->button_pos[2] - KM_BUTTON_DISTANCE_HEIGHT;   /* base re-placed at the recorded cap height */
    }
    if (cfg->rand_objects && ND > N && e->counter == RB_KICK_STEP) {   /* kuka_rand_button_gym_env.py:111-123: applyExternalForce acts on the next stepSimulation */
        double f[3]; int k;
        rb_kick_force(e->obj_xy[9][0], e->obj_xy[9][1], f);
        for (k = 0; k < 3; k++) e->rb_v[10][k] += f[k] * KM_DT / RB_MASS;
    }
    if (action < 0) { if (cfg->action_joints) jt = KM_JOINT_POSITIONS; }      /* None: :295-299, no RNG draw */
    else if (cfg->is_discrete) {
        double dv = KM_DELTA_V + k_normal(r, 0.0, KM_NOISE_STD);
        if (action == 0) motor[0] = -dv; else if (action == 1) motor[0] = dv;
        else if (action == 2) motor[1] = -dv; else if (action == 3) motor[1] = dv;
        else if (action == 4) motor[2] = -dv; else if (action == 5) motor[2] = cfg->force_down ? -dv : dv;
    } else if (cfg->action_joints) {
        double dth = KM_DELTA_THETA + k_normal(r, 0.0, KM_NOISE_STD_JOINTS); int j;
        /* float32 action * python float stays float32, + float64 list -> float64 */
        for (j = 0; j < N; j++) joints[j] = (double)(caction[j] * (float)dth) + KM_JOINT_POSITIONS[j];
        jt = joints;
    } else {
        double dv = KM_DELTA_V_CONTINUOUS + k_normal(r, 0.0, KM_NOISE_STD_CONTINUOUS);
        /* action[i] is a numpy float32 scalar: float32 * python float -> float64 product */
        motor[0] = (double)caction[0] * dv; motor[1] = (double)caction[1] * dv;
        motor[2] = cfg->force_down ? -fabs((double)caction[2] * dv) : (double)caction[2] * dv;
    }
    e->button_motor_on = 1;                                        /* step2 :347 */
    ik_conditioning_reset();
    for (rep = 0; rep < cfg->action_repeat; rep++) {
        physics_step(e, cfg, motor, jt);
        if (termination_cfg(e, cfg)) break;
        e->counter += 1;
    }
    if (t_ik_det < KM_IK_CROSS_DET) e->ik_crossed = 1;
    { double reward = reward_fn(e, cfg); *done = termination_cfg(e, cfg); return reward; }
}

static int g_moving = 0, g_two = 0, g_rand = 0;
/* optional extra traces of the next kuka_oracle_rollout call: q_all [T][n][12] (every DoF), rows [T][n][3] (contact-normal rows, friction rows + 1000 x joint-limit rows,
 * arm <-> free-body contact rows (KukaRandButton) created by the step's last stepSimulation); NULL = off */
static double *g_aux_q = NULL; static int32_t *g_aux_rows = NULL;
void kuka_oracle_set_aux_trace(double *q_all, int32_t *rows) { g_aux_q = q_all; g_aux_rows = rows; }
/* IK conditioning of the next kuka_oracle_rollout call: det [T][n] (smallest det(J^T J + damping I) of the step's IK solves; 1e300 in joint
 * mode), flag [T][n] (the episode's sticky ik_crossed bit after the step, before a possible auto-reset), fin [n][2] (the bit at the end
 * of the rollout, number of env-steps taken with the bit set); NULL = off */
static double *g_aux_ikdet = NULL; static uint8_t *g_aux_ikflag = NULL; static int32_t *g_aux_ikfin = NULL;
void kuka_oracle_set_ik_trace(double *det, uint8_t *flag, int32_t *fin) { g_aux_ikdet = det; g_aux_ikflag = flag; g_aux_ikfin = fin; }
/* KukaRandButton: body state of every env at the end of the next kuka_oracle_rollout call, [n][RB_N][7]: x y z vx vy vz on */
static double *g_aux_bodies = NULL;
void kuka_oracle_set_body_trace(double *bodies) { g_aux_bodies = bodies; }
/* the literal drop of the reference's reset (bodies start 0.1 m / 0.3 m above Z_TABLE, then 500 steps): largest distance of any
 * body of any type from the rest state env_reset places it in */
static void physics_step(kenv *e, const kcfg *cfg, const double motor[5], const double *joint_targets);
double kuka_oracle_rb_drop_check(void) {
    kenv e; kcfg cfg; const double zero[5] = {0, 0, 0, 0, 0}; int i, k; double worst = 0.0;
    memset(&e, 0, sizeof e); memset(&cfg, 0, sizeof cfg);
    cfg.rand_objects = 1; cfg.is_discrete = 1; cfg.action_repeat = 1;
    if (ND <= N) return -1.0;
    for (i = 0; i < ND; i++) e.q[i] = KM_JOINT_POSITIONS[model()->joint_index[i]];
    memcpy(e.ee_target, KM_EE_INIT, sizeof e.ee_target);
    e.button_xy[0] = KM_BUTTON_X; e.button_xy[1] = KM_BUTTON_Y; e.button_z = KM_BUTTON_BASE_Z;
    for (k = 0; k < RB_N; k++) {
        e.rb_on[k] = 1; e.rb_type[k] = k == 10 ? 3 : k % 3;
        e.rb_x[k][0] = 0.3 + 0.04 * k; e.rb_x[k][1] = 0.25; e.rb_x[k][2] = KM_Z_TABLE + (k == 10 ? 0.3 : 0.1);
    }
    for (i = 0; i < KM_N_SETTLE_STEPS; i++) physics_step(&e, &cfg, zero, NULL);
    for (k = 0; k < RB_N; k++) {
        double d = fabs(e.rb_x[k][2] - (model()->table_top_z + rb_height(e.rb_type[k])));
        if (d > worst) worst = d;
        for (i = 0; i < 3; i++) if (fabs(e.rb_v[k][i]) > worst) worst = fabs(e.rb_v[k][i]);
        if (fabs(e.rb_x[k][0] - (0.3 + 0.04 * k)) > worst) worst = fabs(e.rb_x[k][0] - (0.3 + 0.04 * k));
    }
    return worst;
}
/* 0: gripper lumped rigidly into link_7 (7 DoF, rounds 1-2), 1: the full 12-DoF gripper tree with per-link contact spheres and
 * friction rows (kuka_tree_model.h) */
void kuka_oracle_set_full(int full) { g_full = full != 0; model_refresh(); }
int kuka_oracle_get_full(void) { return g_full; }
int kuka_oracle_tree_doubles(void) { return TM_DOUBLES; }
void kuka_oracle_get_tree_model(double *t) { tm_to_table(model(), t); }
void kuka_oracle_set_tree_model(const double *t) { tm_from_table(&g_m, t); g_m_ready = 1; g_full = g_m.nd > 7; inertia_refresh(); }
/* selects KukaMovingButtonGymEnv (1) / Kuka2ButtonGymEnv (2) / KukaRandButtonGymEnv (3) semantics for the following calls
 * (tests are single-threaded callers) */
void kuka_oracle_set_moving(int moving) { g_moving = moving; g_two = 0; g_rand = 0; }
void kuka_oracle_set_variant(int variant) { g_moving = variant == 1; g_two = variant == 2; g_rand = variant == 3; }

/* ------------------------------------------------------------------ batch entry points */
/* Rollout of n envs, T steps each, auto-reset (VecEnv worker semantics).
 * actions: int32 [T][n] (discrete, -1 = None) or float [T][n][adim]; NULL -> Philox random agent.
 * Outputs (any may be NULL): obs0 [n][od], obs [T][n][od] f32, rew f32 / rew64 f64 [T][n], done u8 [T][n],
 * q_trace [T][n][7] f64 (joint positions after each step, BEFORE a possible auto-reset),
 * grip_trace [T][n][3], final [n][40]: q7 qd7 ee3 bq bqd counter n_contacts n_outside terminated button_z,
 * b2q b2qd goal_id n_contacts2 b2x b2y (second button: Kuka2Button only), then q[7..11] qd[7..11] (gripper DoFs, full model). */
int kuka_oracle_rollout(int is_discrete, int action_joints, int random_target, int force_down, int shape_reward,
                        int action_repeat, double max_distance, int obs_mode, int rng_mode, int auto_reset, int n, int T,
                        const int64_t *seeds, const uint32_t *mt_keys, const int32_t *mt_key_len, const void *actions,
                        float *obs0, float *obs, float *rew, double *rew64, uint8_t *done_out, void *act_out,
                        double *q_trace, double *grip_trace, double *final_state, double *ep_stats) {
    kcfg cfg; kenv settled; int e;
    const int od = obs_mode == 1 ? 14 : obs_mode == 2 ? 17 : 3, adim = is_discrete ? 1 : action_joints ? 7 : 3;
    cfg.random_target = random_target; cfg.force_down = force_down; cfg.shape_reward = shape_reward;
    cfg.action_repeat = action_repeat; cfg.is_discrete = is_discrete; cfg.action_joints = action_joints;
    cfg.max_distance = max_distance; cfg.moving = g_moving; cfg.two = g_two; cfg.rand_objects = g_rand;
    settle(&settled, &cfg);
#pragma omp parallel for schedule(dynamic, 4)
    for (e = 0; e < n; e++) {
        kenv env; krng *r = (krng *)malloc(sizeof(krng)); philox_t act; int t;
        double ep_ret = 0, last_ret = 0; int ep_len = 0, last_len = 0, n_fin = 0, n_ikx = 0;
        r->mode = rng_mode;
        if (rng_mode == 2) np_rng_seed_array(&r->mt, mt_keys + 2 * (size_t)e, mt_key_len[e]);
        r->ph.k0 = (uint32_t)(uint64_t)seeds[e]; r->ph.k1 = (uint32_t)((uint64_t)seeds[e] >> 32); r->ph.ctr = 0; r->ph.stream = 0;
        act = r->ph; act.stream = 1;
        env_reset(&env, &cfg, r, &settled);
        if (obs0) observe(&env, obs_mode, obs0 + (size_t)e * od);
        for (t = 0; t < T; t++) {
            size_t row = (size_t)t * n + e; int a = 0, done; float ca[7] = {0}; double reward;
            if (actions) {
                if (is_discrete) a = ((const int32_t *)actions)[row];
                else memcpy(ca, (const float *)actions + row * adim, sizeof(float) * adim);
            } else {
                if (is_discrete) a = (int)philox_bounded(&act, 5);
                else { int j; for (j = 0; j < adim; j += 2) { uint32_t o[4]; philox_block(&act, o); ca[j] = (float)(-1.0 + 2.0 * philox_to_double(o[0], o[1])); if (j + 1 < adim) ca[j + 1] = (float)(-1.0 + 2.0 * philox_to_double(o[2], o[3])); } }
                if (act_out) { if (is_discrete) ((int32_t *)act_out)[row] = a; else memcpy((float *)act_out + row * adim, ca, sizeof(float) * adim); }
            }
            reward = env_step(&env, &cfg, r, a, ca, &done);
            if (q_trace) memcpy(q_trace + row * N, env.q, sizeof(double) * N);
            if (g_aux_q) { memset(g_aux_q + row * TN, 0, sizeof(double) * TN); memcpy(g_aux_q + row * TN, env.q, sizeof(double) * ND); }
            if (g_aux_rows) { g_aux_rows[3 * row] = g_probe_rows[0]; g_aux_rows[3 * row + 1] = g_probe_rows[1] + 1000 * g_probe_rows[2]; g_aux_rows[3 * row + 2] = g_probe_rows[3]; }
            if (grip_trace) memcpy(grip_trace + row * 3, env.gripper, sizeof(double) * 3);
            if (g_aux_ikdet) g_aux_ikdet[row] = t_ik_det;
            if (g_aux_ikflag) g_aux_ikflag[row] = (uint8_t)env.ik_crossed;
            n_ikx += env.ik_crossed;
            ep_ret += reward; ep_len += 1;
            if (done) {
                last_ret = ep_ret; last_len = ep_len; n_fin += 1; ep_ret = 0; ep_len = 0;
                if (auto_reset) env_reset(&env, &cfg, r, &settled);
            }
            if (obs) observe(&env, obs_mode, obs + row * od);
            if (rew) rew[row] = (float)reward;
            if (rew64) rew64[row] = reward;
            if (done_out) done_out[row] = (uint8_t)done;
        }
        if (final_state) {
            double *f = final_state + 40 * (size_t)e; int j;
            for (j = 0; j < 5; j++) { f[30 + j] = N + j < ND ? env.q[N + j] : 0.0; f[35 + j] = N + j < ND ? env.qd[N + j] : 0.0; }
            for (j = 0; j < N; j++) { f[j] = env.q[j]; f[7 + j] = env.qd[j]; }
            f[14] = env.ee_target[0]; f[15] = env.ee_target[1]; f[16] = env.ee_target[2]; f[17] = env.bq; f[18] = env.bqd;
            f[19] = env.counter; f[20] = env.n_contacts; f[21] = env.n_outside; f[22] = env.terminated; f[23] = env.button_pos[2];
            if (cfg.moving) f[23] = env.button_pos[1];
            f[24] = env.b2q; f[25] = env.b2qd; f[26] = env.goal_id; f[27] = env.n_contacts2; f[28] = env.button2_xy[0]; f[29] = env.button2_xy[1];
        }
        if (g_aux_bodies) {
            int k, j; double *bd = g_aux_bodies + (size_t)e * RB_N * 7;
            for (k = 0; k < RB_N; k++) { for (j = 0; j < 3; j++) { bd[7 * k + j] = env.rb_x[k][j]; bd[7 * k + 3 + j] = env.rb_v[k][j]; } bd[7 * k + 6] = env.rb_on[k]; }
        }
        if (g_aux_ikfin) { g_aux_ikfin[2 * (size_t)e] = env.ik_crossed; g_aux_ikfin[2 * (size_t)e + 1] = n_ikx; }
        if (ep_stats) { ep_stats[3 * (size_t)e] = last_ret; ep_stats[3 * (size_t)e + 1] = last_len; ep_stats[3 * (size_t)e + 2] = n_fin; }
        margin_merge();
        free(r);
    }
    return 0;
}

/* The settled state itself (q7 qd7 ee3 bq bqd gripper3) and raw dynamics probes for the
 * independent numpy cross-check (tests/test_kuka_dynamics.py). */
void kuka_oracle_settled(int random_target, int action_joints, double *out22) {
    kcfg cfg; kenv s; int j;
    memset(&cfg, 0, sizeof cfg); cfg.random_target = random_target; cfg.action_joints = action_joints; cfg.action_repeat = 1; cfg.is_discrete = 1; cfg.two = g_two;
    settle(&s, &cfg);
    for (j = 0; j < N; j++) { out22[j] = s.q[j]; out22[7 + j] = s.qd[j]; }
    out22[14] = s.ee_target[0]; out22[15] = s.ee_target[1]; out22[16] = s.ee_target[2]; out22[17] = s.bq; out22[18] = s.bqd;
    out22[19] = s.gripper[0]; out22[20] = s.gripper[1]; out22[21] = s.gripper[2];
}
void kuka_oracle_aba(const double *q, const double *qd, const double *tau, double gz, double *qdd) { aba(q, qd, tau, gz, qdd); }
void kuka_oracle_minv(const double *q, double *Wnn) { double W[TN][TN]; int i, j; const int n = ND; mass_matrix_inverse(q, W); for (i = 0; i < n; i++) for (j = 0; j < n; j++) Wnn[i * n + j] = W[i][j]; }
void kuka_oracle_fk(const double *q, double *R63, double *p21) {
    mat3 R[TN]; double p[TN][3]; int i, a, b;
    forward_kinematics(q, R, p);
    for (i = 0; i < ND; i++) { for (a = 0; a < 3; a++) { for (b = 0; b < 3; b++) R63[i * 9 + a * 3 + b] = R[i][a][b]; p21[i * 3 + a] = p[i][a]; } }
}
void kuka_oracle_ik(const double *q, const double *target, double *q_des) {
    mat3 R[TN]; double p[TN][3]; forward_kinematics(q, R, p); inverse_kinematics(q, R, p, target, g_two ? KM_IK_DAMPING_DEFAULT : KM_IK_DAMPING, q_des);
}
/* scripted-wrapper probe: reward/termination logic on caller-supplied physics outputs */
void kuka_oracle_wrapper_step(double *state8, const double *gripper, const double *button_pos, int contact_button, int contact_table,
                              int shape_reward, int is_discrete, double max_distance, double *reward, int *done) {
    kenv e; kcfg cfg; memset(&e, 0, sizeof e); memset(&cfg, 0, sizeof cfg);
    e.counter = (int)state8[0]; e.n_contacts = (int)state8[1]; e.n_outside = (int)state8[2]; e.terminated = (int)state8[3];
    memcpy(e.gripper, gripper, sizeof e.gripper); memcpy(e.button_pos, button_pos, sizeof e.button_pos);
    e.contact_button = contact_button; e.contact_table = contact_table;
    cfg.shape_reward = shape_reward; cfg.is_discrete = is_discrete; cfg.max_distance = max_distance; cfg.moving = g_moving;
    if (!termination_cfg(&e, &cfg)) e.counter += 1;                  /* step2 loop with action_repeat = 1 */
    *reward = reward_fn(&e, &cfg); *done = termination_cfg(&e, &cfg);
    state8[0] = e.counter; state8[1] = e.n_contacts; state8[2] = e.n_outside; state8[3] = e.terminated;
}
/* same probe for Kuka2ButtonGymEnv: state8 = counter n_contacts[0] n_outside terminated goal_id n_contacts[1];
 * contact_goal = contact with the CURRENT goal button (any link) */
void kuka_oracle_wrapper_step_two(double *state8, const double *gripper, const double *all_pos6, int contact_goal, int contact_table,
                                  int shape_reward, double max_distance, double *reward, int *done) {
    kenv e; kcfg cfg; memset(&e, 0, sizeof e); memset(&cfg, 0, sizeof cfg);
    e.counter = (int)state8[0]; e.n_contacts = (int)state8[1]; e.n_outside = (int)state8[2]; e.terminated = (int)state8[3];
    e.goal_id = (int)state8[4]; e.n_contacts2 = (int)state8[5];
    memcpy(e.gripper, gripper, sizeof e.gripper); memcpy(e.all_pos, all_pos6, sizeof e.all_pos);
    e.contact_body[e.goal_id] = contact_goal; e.contact_table = contact_table;
    cfg.shape_reward = shape_reward; cfg.is_discrete = 1; cfg.max_distance = max_distance; cfg.two = 1;
    if (!termination_cfg(&e, &cfg)) e.counter += 1;
    *reward = reward_fn(&e, &cfg); *done = termination_cfg(&e, &cfg);
    state8[0] = e.counter; state8[1] = e.n_contacts; state8[2] = e.n_outside; state8[3] = e.terminated;
    state8[4] = e.goal_id; state8[5] = e.n_contacts2;
}

static double g_last_buttons[4], g_last_objects[30];
/* KukaRandButton: (x, y, present) of the ten distractors drawn by the reset() of the last kuka_oracle_command_trace call */
void kuka_oracle_last_objects(double *out30) { memcpy(out30, g_last_objects, sizeof g_last_objects); }
/* button base positions drawn by the reset() of the last kuka_oracle_command_trace call: b1x b1y b2x b2y */
void kuka_oracle_last_buttons(double *out4) { memcpy(out4, g_last_buttons, sizeof g_last_buttons); }

/* Commands issued by reset() (5 init actions) and by every step of one env: Cartesian IK targets
 * (Kuka.end_effector_pos after the clip) and, in joint mode, the motor target list.  Returns the number of
 * env steps executed (stops at done). */
int kuka_oracle_command_trace(int is_discrete, int action_joints, int random_target, int force_down, const uint32_t *mt_key,
                              int mt_key_len, int T, const void *actions, double *ee_trace, double *jt_trace, int *n_reset_cmds) {
    kcfg cfg; kenv settled, env; krng *r = (krng *)malloc(sizeof(krng)); int t, done = 0;
    cfg.random_target = random_target; cfg.force_down = force_down; cfg.shape_reward = 0; cfg.action_repeat = 1;
    cfg.is_discrete = is_discrete; cfg.action_joints = action_joints; cfg.max_distance = g_two ? 2.0 : 0.8; cfg.moving = g_moving; cfg.two = g_two; cfg.rand_objects = g_rand;
    settle(&settled, &cfg);
    r->mode = 2; np_rng_seed_array(&r->mt, mt_key, mt_key_len);
    g_trace_ee = ee_trace; g_trace_jt = jt_trace; g_trace_n = 0;
    env_reset(&env, &cfg, r, &settled);
    *n_reset_cmds = g_trace_n;
    { int k; for (k = 0; k < 10; k++) { g_last_objects[3 * k] = env.obj_xy[k][0]; g_last_objects[3 * k + 1] = env.obj_xy[k][1]; g_last_objects[3 * k + 2] = env.obj_present[k]; } }
    g_last_buttons[0] = env.button_xy[0]; g_last_buttons[1] = env.button_xy[1]; g_last_buttons[2] = env.button2_xy[0]; g_last_buttons[3] = env.button2_xy[1];
    for (t = 0; t < T && !done; t++) {
        int a = 0; float ca[7] = {0};
        if (is_discrete) a = ((const int32_t *)actions)[t];
        else memcpy(ca, (const float *)actions + (size_t)t * (action_joints ? 7 : 3), sizeof(float) * (action_joints ? 7 : 3));
        env_step(&env, &cfg, r, a, ca, &done);
    }
    g_trace_ee = NULL; g_trace_jt = NULL;
    free(r);
    return t;
}

/* ------------------------------------------------------------------ single-env handle
 * One KukaButtonGymEnv object the way a SubprocVecEnv worker holds it (rl_baselines/utils.py:216-220,
 * environments/utils.py:48-57): bench.py's cpu_baseline leg drives one of these per worker process through
 * oracle/subproc_baseline.py.  reset() integrates the literal 505 steps (kuka_button_gym_env.py:242-269). */
typedef struct { kcfg cfg; kenv env; krng rng; int obs_mode; } koracle_env;

void *kuka_oracle_env_new(int is_discrete, int random_target, int force_down, int shape_reward, int action_repeat,
                          double max_distance, int obs_mode, int rng_mode, int64_t seed, const uint32_t *mt_key, int mt_key_len) {
    koracle_env *h = (koracle_env *)calloc(1, sizeof(koracle_env));
    h->cfg.is_discrete = is_discrete; h->cfg.random_target = random_target; h->cfg.force_down = force_down;
    h->cfg.shape_reward = shape_reward; h->cfg.action_repeat = action_repeat; h->cfg.max_distance = max_distance;
    h->cfg.moving = g_moving; h->cfg.two = g_two; h->cfg.rand_objects = g_rand;
    h->obs_mode = obs_mode; h->rng.mode = rng_mode;
    if (rng_mode == 2) np_rng_seed_array(&h->rng.mt, mt_key, mt_key_len);
    h->rng.ph.k0 = (uint32_t)(uint64_t)seed; h->rng.ph.k1 = (uint32_t)((uint64_t)seed >> 32); h->rng.ph.ctr = 0; h->rng.ph.stream = 0;
    return h;
}
void kuka_oracle_env_reset(void *hv, float *obs) {
    koracle_env *h = (koracle_env *)hv; kenv settled;
    settle(&settled, &h->cfg);                        /* the reference re-runs its 500 settle steps on every reset */
    env_reset(&h->env, &h->cfg, &h->rng, &settled);
    observe(&h->env, h->obs_mode, obs);
}
double kuka_oracle_env_step(void *hv, int action, float *obs, int *done) {
    koracle_env *h = (koracle_env *)hv;
    const float ca[7] = {0};
    double r = env_step(&h->env, &h->cfg, &h->rng, action, ca, done);
    observe(&h->env, h->obs_mode, obs);
    return r;
}
void kuka_oracle_env_free(void *hv) { free(hv); }

/* runtime model table (oracle/kuka_model.h): this translation unit's copy */
void kuka_oracle_set_model(const double *table138) { km_set_model(table138); model_refresh(); }
void kuka_oracle_get_model(double *table138) { km_get_model(table138); }
