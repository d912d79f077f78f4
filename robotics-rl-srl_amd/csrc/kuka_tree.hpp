// kuka_tree.hpp — lane-group physics step for the FULL Kuka model (kuka_tree_model.hpp): the 12-DoF arm + gripper tree with every
// joint motor of kuka.py:167-187, per-link contact spheres and one friction row per contact.  Same decomposition as
// kuka_group.hpp (16 lanes = one DPP row per env, broadcast-accumulate for every cross-lane pattern), generalised from a chain to
// a kinematic tree:
//
//   lane l < 12 owns DoF / link l and motor row l; lanes 12, 13, 14 own the button's scalar rows (motor, lower stop, upper stop);
//   every lane owns one collision sphere (16) and — on the rare steps that carry them — one row of a second row bank "B":
//   slots 0..7 joint-limit and contact-normal rows in creation order, slot 8 + g the friction row of contact-normal slot g.
//
//   * kinematics: local joint transforms composed by 4 levels of pointer jumping over the tree (lane l reads lane parent^(2^k)(l));
//   * dynamics in WORLD coordinates about the world origin: link velocities / bias accelerations are sums over ANCESTORS, link
//     forces / composite inertias sums over DESCENDANTS — masked row-broadcast FMAs with per-lane ancestor / descendant masks;
//     M by CRBA (M_kl = S_k . Ic_l S_l for k an ancestor of l, 0 across branches), M^-1 by the lane-parallel Gauss-Jordan sweep;
//   * IK: the end effector (link 6) does not move with the gripper joints, so the damped-least-squares step is the 7x7 arm block;
//   * projected Gauss-Seidel in impulse space: bank-A rows (motors, button) scaled to [0, 1] and updated by the fused clamp /
//     broadcast-FMA row of kuka_group.hpp; bank-B rows in impulse units with explicit bounds (a friction row's bounds follow its
//     normal row's current impulse, btMultiBodyConstraintSolver::solveSingleIteration); their coupling coefficients live in LDS.
//
// The same source runs on the host under the fiber harness (csrc/kuka_hostcheck.cpp); the oracle it is compared with is
// oracle/kuka_oracle.c in its full-model mode (link-frame ABA, dv-space Gauss-Seidel): ~1e-11 on joints, flags bit for bit.
#pragma once
#include "kuka_group.hpp"
#include "kuka_tree_model.hpp"

namespace srl {
namespace kuka {
namespace tree {
using grp::GL;
using grp::GState;
using grp::bcast;
using grp::ballot;
using grp::clamp01;
using grp::compose;
using grp::fmac_bcast;
using grp::gany;
using grp::lane_id;
using grp::pgs_row;
using grp::pgs_row2;
using grp::rcp;
using grp::shfl;
using grp::sync_scratch;
using grp::wany;

// ---- phase stamps of the PROFILING build (make prof: -DSRL_TREE_PROF, profiles/probes/kuka_tree_phases.py): shader-clock cycles
// between consecutive stamps, accumulated per phase by lane 0 of the workgroup in LDS, printed by workgroup 0 when the kernel ends.
// The product build compiles them away.
#if defined(SRL_TREE_PROF) && SRL_G_DEVICE
constexpr int kProfSlots = 22;      // 0..11: phases of a step; 12..16: inside general_path; 17..19: number of steps with generic rows / a limit row / a contact row; 20, 21: loop top -> action sampled -> command mapped
SRL_G unsigned long long *tprof_buf() { __shared__ unsigned long long p[kProfSlots + 1]; return p; }
#define SRL_TCOUNT(i) do { if (threadIdx.x == 0) tprof_buf()[i] += 1; } while (0)
#define SRL_TSTAMP(i) do { if (threadIdx.x == 0) { unsigned long long *p_ = tprof_buf(); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); p_[i] += t_ - p_[kProfSlots]; p_[kProfSlots] = t_; } } while (0)
#else
#define SRL_TSTAMP(i)
#define SRL_TCOUNT(i)
#endif

constexpr int NJ = 12;                          // joint lanes
constexpr int NA = 7;                           // arm joints (IK, commands)
constexpr int kBM = 12, kBLo = 13, kBHi = 14;   // lanes of the button's scalar rows
constexpr int kNGen = 6, kNB = 2 * kNGen;       // bank-B slots; slots < kNGen: limits + contact normals, kNGen + g: friction of normal g
constexpr int kNGen2 = 4;                       // SRLHIP_KUKA_DETAIL_FRICTION2: the same 12 slots as 4 normals / limits + 4 + 4 friction rows (slot ng + g, 2 ng + g)
constexpr int kNArows = 15;                     // bank-A rows: 12 motors + the button's three
constexpr int kTreeStartDoubles = 4 * NJ + 8;   // q12 qd12 sq12 cq12 ee3 bq bqd grip3
// LDS scratch per env (doubles): row definitions J[16][12], W J [16][12], then three coupling planes [row j][lane i]:
// nBA (lane i's B row <- A row j), nAB (lane i's A row <- B row j), nBB (lane i's B row <- B row j)
// and the row scalars DEF[slot][8]: Jb, desired velocity, position-error velocity, upper bound, on, mu
constexpr int kDefDoubles = 8;
constexpr int SC_J = 0, SC_WJ = kNB * NJ, SC_NBA = 2 * kNB * NJ, SC_NAB = SC_NBA + kNArows * GL, SC_NBB = SC_NAB + kNB * GL, SC_DEF = SC_NBB + kNB * GL,
              SC_S = SC_DEF + kNB * kDefDoubles;                 // + the joints' spatial axes S[12][6] (contact Jacobians)
constexpr int kTreeScratchDoubles = SC_S + NJ * 6;               // 1080 doubles = 8.4 KiB per env
// OCC = 1 (two wavefronts per SIMD, kuka_tree_occ.hip): ONE work area per wavefront (everything up to SC_S: the row definitions and
// coupling planes of the general path, taken by the wavefront's four envs in turns) + a small per-env park (the own row of M^-1 and
// the spatial axes: what a later turn still needs of the step's dynamics); sphere / limit candidates are recomputed in the turn
// instead of being parked.
constexpr int kTreeWorkDoubles = SC_S;                           // 1008 doubles = 7.9 KiB per wavefront
constexpr int PK_W = 0, PK_S = NJ * GL, kTreeParkDoubles = PK_S + NJ * 6;     // 264 doubles = 2.1 KiB per env

// ------------------------------------------------------------------ KukaRandButtonGymEnv free bodies (template flag RB)
// kuka_rand_button_gym_env.py:59-71 drops ten meshes and a ball on the table, :111-125 kicks the ball at env step 10: free bodies the
// arm can push.  Same restatement as oracle/kuka_oracle.c (free-body section: proxy shapes, masses, what is not modelled): lane
// k < 11 owns body k (0..9 the distractors in draw order, 10 the ball) — its 3 translational DoFs, its table-contact row and two
// table-friction rows; arm-sphere <-> body contacts are bank-B rows that also act on the body's velocity change (kept on the
// owning lane, dv space).  A body the arm does not touch is solved in closed form (its three rows are orthogonal and decoupled:
// the first sweep is the fixed point of all 150); with an arm contact the body's rows are swept with the arm's.
constexpr int kRbN = 11, kRbKickStep = 10;
constexpr double kRbBallR = 0.03, kRbMass = 0.1, kRbMuTable = 0.25, kRbBallForce = 10.0, kRbMaxHeight = 0.07;
struct RBody {
    double x[3], v[3];        // centre of the proxy shape, velocity
    double ox, oy;            // the reset draws of the distractor (type hash; lane 9's feed the kick direction)
    int type; bool on;        // 0 duck / 1 lego / 2 cube (boxes), 3 ball; lanes >= 11 and dropped candidates: on = false
};
SRL_G double rb_half(int type, int axis) {
    return type == 0 ? (axis == 0 ? 0.05 : 0.035) : type == 1 ? (axis == 0 ? 0.016 : axis == 1 ? 0.032 : 0.012) : 0.025;
}
SRL_G double rb_height(int type) { return type == 3 ? kRbBallR : rb_half(type, 2); }
SRL_G int rb_type_of(double ox, double oy) {       // the reference's type comes from the unseeded global RNG: a hash of the drawn position (as the renderer)
    unsigned long long bx, by;
    memcpy(&bx, &ox, 8); memcpy(&by, &oy, 8);
    return (int)((uint32_t)((bx >> 20) ^ (by >> 20)) % 3u);
}
SRL_G void rb_kick_force(double ox, double oy, double f[3]) {
#pragma clang fp contract(off)
    unsigned long long bx, by;
    memcpy(&bx, &ox, 8); memcpy(&by, &oy, 8);
    const unsigned long long h = (bx >> 12) ^ (by >> 20) ^ (bx >> 36);
    const double u = (double)(h & 0xFFFFFFull) / 16777216.0;
    const double a = 1.0 - u, b = u, len = sqrt(a * a + b * b);
    f[0] = kRbBallForce * (a / len); f[1] = kRbBallForce * (b / len); f[2] = 1.0;
}
// signed distance between a sphere and a body's shape; n = unit normal from the body towards the sphere
SRL_G double sphere_body(const double c[3], double rad, const double x[3], int type, double n[3]) {
#pragma clang fp contract(off)
    const double d[3] = {c[0] - x[0], c[1] - x[1], c[2] - x[2]};
    if (type == 3) {
        const double len = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        if (len > 1e-12) { n[0] = d[0] / len; n[1] = d[1] / len; n[2] = d[2] / len; } else { n[0] = 0; n[1] = 0; n[2] = 1; }
        return len - kRbBallR - rad;
    }
    double q[3], diff[3];
    bool inside = true;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const double h = rb_half(type, k);
        q[k] = d[k] < -h ? -h : (d[k] > h ? h : d[k]); diff[k] = d[k] - q[k];
        if (diff[k] != 0.0) inside = false;
    }
    if (inside) {                                   // centre inside the box: exit through the nearest face
        int best = 0; double pen = rb_half(type, 0) - fabs(d[0]);
#pragma unroll
        for (int k = 1; k < 3; k++) { const double pk = rb_half(type, k) - fabs(d[k]); if (pk < pen) { pen = pk; best = k; } }
#pragma unroll
        for (int k = 0; k < 3; k++) n[k] = k == best ? (d[k] < 0 ? -1.0 : 1.0) : 0.0;
        return -pen - rad;
    }
    const double len = sqrt(diff[0] * diff[0] + diff[1] * diff[1] + diff[2] * diff[2]);
    n[0] = diff[0] / len; n[1] = diff[1] / len; n[2] = diff[2] / len;
    return len - rad;
}
// the body's own rows: table normal (0, 0, 1), friction along btPlaneSpace1's tangent (0, -1, 0) and n x t1 = (1, 0, 0); right-hand
// sides already times the row's 1 / D = the body's mass (every Jacobian is a unit vector acting on the one body)
struct RbRows { bool hasT, fric; double rhsN, rhsT1, rhsT2, lamN, lamT1, lamT2, dv[3]; };
SRL_G void rb_rows_setup(const RBody &B, double table_z, double cerp, double slop, bool friction, RbRows &r) {
#pragma clang fp contract(off)
    const double inv_dt = 1.0 / kDt, Dinv = 1.0 / (1.0 / kRbMass);
    const double dist = B.x[2] - rb_height(B.type) - table_z, pen = dist + slop;
    r.hasT = B.on && dist < kContactThreshold; r.fric = friction;
    r.rhsN = ((pen > 0 ? -pen * inv_dt : 0.0) - B.v[2]) * Dinv + (pen > 0 ? 0.0 : -pen * cerp * inv_dt) * Dinv;
    r.rhsT1 = (0.0 - (-B.v[1])) * Dinv; r.rhsT2 = (0.0 - B.v[0]) * Dinv;
    r.lamN = 0.0; r.lamT1 = 0.0; r.lamT2 = 0.0; r.dv[0] = 0.0; r.dv[1] = 0.0; r.dv[2] = 0.0;
}
SRL_G void rb_sweep_normal(RbRows &r) {
#pragma clang fp contract(off)
    if (!r.hasT) return;
    const double Dinv = 1.0 / (1.0 / kRbMass);
    double delta = r.rhsN - r.dv[2] * Dinv;
    const double sum = r.lamN + delta;
    if (sum < 0.0) { delta = 0.0 - r.lamN; r.lamN = 0.0; } else if (sum > 1e10) { delta = 1e10 - r.lamN; r.lamN = 1e10; } else r.lamN = sum;
    r.dv[2] += delta * 1.0 / kRbMass;
}
SRL_G void rb_sweep_friction(RbRows &r) {
#pragma clang fp contract(off)
    if (!r.hasT || !r.fric || !(r.lamN > 0.0)) return;
    const double Dinv = 1.0 / (1.0 / kRbMass), lo = -kRbMuTable * r.lamN, hi = kRbMuTable * r.lamN;
    {   // t1 = (0, -1, 0)
        double delta = r.rhsT1 - (-r.dv[1]) * Dinv;
        const double sum = r.lamT1 + delta;
        if (sum < lo) { delta = lo - r.lamT1; r.lamT1 = lo; } else if (sum > hi) { delta = hi - r.lamT1; r.lamT1 = hi; } else r.lamT1 = sum;
        r.dv[1] += delta * -1.0 / kRbMass;
    }
    {   // t2 = (1, 0, 0)
        double delta = r.rhsT2 - r.dv[0] * Dinv;
        const double sum = r.lamT2 + delta;
        if (sum < lo) { delta = lo - r.lamT2; r.lamT2 = lo; } else if (sum > hi) { delta = hi - r.lamT2; r.lamT2 = hi; } else r.lamT2 = sum;
        r.dv[0] += delta * 1.0 / kRbMass;
    }
}

// ------------------------------------------------------------------ per-lane constants
struct TLane {
    int l;
    bool jnt, arm;            // l < 12, l < 7
    double jm, am;            // 1.0 on joint / arm lanes
    double e[NJ];             // e[j] = (l == j)
    uint32_t anc, desc;       // bit k: DoF k is an ancestor-or-self / descendant-or-self of the own link
    int src[4];               // lane parent^(1, 2, 4, 8)(l); beyond the root: lane 15 (identity)
    double mass, mcomp;       // link mass, mass of the own subtree
    double com[3], in[6];     // centre of mass, inertia about it (xx xy xz yy yz zz) in link axes
    double F[9], t[3], ax[3]; // fixed rotation (columns) and origin of the joint frame in the parent link, joint axis in the joint frame
    double jlo, jhi;          // joint limits (jlo > jhi: none)
    double damping, kp, bound, maxvel, q0;   // motor: gain, impulse bound force * dt, velocity clamp; kJointPositions of the joint
    double tsel;              // gripper motor target = tsel * finger_angle (-1 joint 8, +1 joint 11, else 0)
    int slink; uint32_t sanc; // own collision sphere: link, that link's ancestor mask; slink < 0: none
    double sph[4], smu;
    int ee_link, grip_link, max_gen;
    bool friction;
    double eept[3], grpt[3], table_z, base_z;
    int detail, ng;           // solver_detail bits; normal / limit slots of bank B in force (6, or 4 with two friction directions)
    double cerp, lerp, slop;  // contact_erp, limit_erp, linear_slop
};

SRL_G void lane_init(TLane &L, const TreeModel *m) {
    const int l = lane_id();
    const int nd = (int)m->nd;
    L.l = l; L.jnt = l < nd; L.arm = l < NA; L.jm = L.jnt ? 1.0 : 0.0; L.am = L.arm ? 1.0 : 0.0;
#pragma unroll
    for (int j = 0; j < NJ; j++) L.e[j] = l == j ? 1.0 : 0.0;
    const int i = L.jnt ? l : 0;
    const TreeJoint &J = m->j[i];
    // ancestors of the own link, descendants, pointer-jumping sources
    L.anc = 0; L.desc = 0;
    if (L.jnt) for (int k = l; k >= 0; k = (int)m->j[k].parent) L.anc |= 1u << k;
    double mc = 0.0;
    for (int k = 0; k < nd; k++) {
        bool under = false;
        for (int a = k; a >= 0; a = (int)m->j[a].parent) under = under || a == l;
        if (under && L.jnt) { L.desc |= 1u << k; mc += m->j[k].mass; }
    }
    {
        int a = L.jnt ? (int)J.parent : -1, hop = 1;
        for (int lvl = 0; lvl < 4; lvl++) {
            L.src[lvl] = a >= 0 ? a : GL - 1;      // beyond the root: lane 15, which never owns a joint and carries the identity
            // the next source is 2 * hop links up from l
            for (int s = 0; s < hop && a >= 0; s++) a = (int)m->j[a].parent;
            hop *= 2;
        }
    }
    L.mass = L.jnt ? J.mass : 0.0; L.mcomp = L.jnt ? mc : 0.0;
    for (int k = 0; k < 3; k++) { L.com[k] = J.com[k]; L.t[k] = L.jnt ? J.xyz[k] + (J.parent < 0 ? kBasePos[k] : 0.0) : 0.0; L.ax[k] = J.axis[k]; }
    for (int k = 0; k < 6; k++) L.in[k] = L.jnt ? J.inertia[k] : 0.0;
    for (int c = 0; c < 3; c++) for (int r = 0; r < 3; r++) L.F[3 * c + r] = L.jnt ? J.Rj[3 * r + c] : (r == c ? 1.0 : 0.0);
    L.jlo = L.jnt ? J.lower : 1.0; L.jhi = L.jnt ? J.upper : -1.0;
    L.damping = L.jnt ? J.damping : 0.0; L.kp = J.kp; L.bound = L.jnt ? J.max_force * kDt : 0.0; L.maxvel = J.max_vel;
    L.q0 = L.jnt ? kJointPositions[(int)J.joint_index] : 0.0;
    L.tsel = !L.jnt ? 0.0 : (int)J.joint_index == 8 ? -1.0 : (int)J.joint_index == 11 ? 1.0 : 0.0;
    const int ns = (int)m->nsphere;
    const TreeSphere &S = m->s[l < ns ? l : 0];
    L.slink = l < ns ? (int)S.link : -1;
    L.sanc = 0;
    if (L.slink >= 0) for (int k = L.slink; k >= 0; k = (int)m->j[k].parent) L.sanc |= 1u << k;
    for (int k = 0; k < 3; k++) L.sph[k] = S.c[k];
    L.sph[3] = S.r; L.smu = S.mu;
    L.ee_link = (int)m->ee_link; L.grip_link = (int)m->grip_link; L.max_gen = (int)m->max_generic_rows; L.friction = m->friction != 0.0;
    L.detail = (int)m->solver_detail; L.ng = (L.detail & kDetailFriction2) ? kNGen2 : kNGen;
    if (L.max_gen > L.ng) L.max_gen = L.ng;
    L.cerp = m->contact_erp; L.lerp = m->limit_erp; L.slop = m->linear_slop;
    for (int k = 0; k < 3; k++) { L.eept[k] = m->ee_point[k]; L.grpt[k] = m->grip_point[k]; }
    L.table_z = m->table_top_z; L.base_z = m->button_base_z;
}

// The per-lane constants live in LDS for the whole launch ([field][16 lanes], one table per wavefront: its envs share the model)
// and are read where a physics step uses them (TL below): kept in registers across the rollout loop (~75 doubles per lane) they
// pushed the loop into scratch, and any scratch reload inside the step loop waits for the previous step's output stores (gfx9
// counts loads and stores in one in-order counter).  LDS reads count on lgkmcnt: ~140 ds_read_b64 per step.
enum { LT_MASS = 0, LT_MCOMP, LT_COM, LT_IN = LT_COM + 3, LT_F = LT_IN + 6, LT_T = LT_F + 9, LT_AX = LT_T + 3, LT_JLO = LT_AX + 3, LT_JHI, LT_DAMP, LT_KP,
       LT_BOUND, LT_MAXVEL, LT_Q0, LT_TSEL, LT_SPH, LT_SMU = LT_SPH + 4, LT_ANC, LT_DESC, LT_SRC, LT_SLINK = LT_SRC + 4, LT_SANC, LT_COUNT };
// behind the per-lane fields: the model's scalars, stored once
enum { LS_EEPT = 0, LS_GRPT = 3, LS_TABLEZ = 6, LS_BASEZ, LS_EELINK, LS_GRIPLINK, LS_MAXGEN, LS_FRICTION, LS_DETAIL, LS_NG, LS_CERP, LS_LERP, LS_SLOP, LS_COUNT };
constexpr int kLaneTableDoubles = LT_COUNT * GL + LS_COUNT;
SRL_G void lane_store(const TLane &L, double *tab) {
    double *t = tab + L.l;
#define SRL_PUT(F, v) t[(F) * GL] = (double)(v);
    SRL_PUT(LT_MASS, L.mass) SRL_PUT(LT_MCOMP, L.mcomp)
#pragma unroll
    for (int k = 0; k < 3; k++) { SRL_PUT(LT_COM + k, L.com[k]) SRL_PUT(LT_T + k, L.t[k]) SRL_PUT(LT_AX + k, L.ax[k]) }
#pragma unroll
    for (int k = 0; k < 6; k++) SRL_PUT(LT_IN + k, L.in[k])
#pragma unroll
    for (int k = 0; k < 9; k++) SRL_PUT(LT_F + k, L.F[k])
#pragma unroll
    for (int k = 0; k < 4; k++) { SRL_PUT(LT_SPH + k, L.sph[k]) SRL_PUT(LT_SRC + k, L.src[k]) }
    SRL_PUT(LT_JLO, L.jlo) SRL_PUT(LT_JHI, L.jhi) SRL_PUT(LT_DAMP, L.damping) SRL_PUT(LT_KP, L.kp) SRL_PUT(LT_BOUND, L.bound) SRL_PUT(LT_MAXVEL, L.maxvel)
    SRL_PUT(LT_Q0, L.q0) SRL_PUT(LT_TSEL, L.tsel) SRL_PUT(LT_SMU, L.smu)
    SRL_PUT(LT_ANC, L.anc) SRL_PUT(LT_DESC, L.desc) SRL_PUT(LT_SLINK, L.slink) SRL_PUT(LT_SANC, L.sanc)
#undef SRL_PUT
    if (L.l == 0) {
        double *sc = tab + LT_COUNT * GL;
#pragma unroll
        for (int k = 0; k < 3; k++) { sc[LS_EEPT + k] = L.eept[k]; sc[LS_GRPT + k] = L.grpt[k]; }
        sc[LS_TABLEZ] = L.table_z; sc[LS_BASEZ] = L.base_z; sc[LS_EELINK] = L.ee_link; sc[LS_GRIPLINK] = L.grip_link;
        sc[LS_MAXGEN] = L.max_gen; sc[LS_FRICTION] = L.friction ? 1.0 : 0.0;
        sc[LS_DETAIL] = L.detail; sc[LS_NG] = L.ng; sc[LS_CERP] = L.cerp; sc[LS_LERP] = L.lerp; sc[LS_SLOP] = L.slop;
    }
}
// Lazy view of the lane table: every field is read from LDS where it is used (loading all ~45 of them at the top of a step kept
// ~100 registers live through the dynamics).  The pointer is laundered per view, so nothing is hoisted out of the rollout loop.
struct TL {
    const double *p;          // table + own lane
    const double *sc;         // the model's scalars behind the per-lane fields
    int l;
    bool jnt, arm;
    double jm, am;
    SRL_G double f(int F) const { return p[F * GL]; }
    SRL_G double e(int j) const { return l == j ? 1.0 : 0.0; }
    SRL_G double mass() const { return f(LT_MASS); }
    SRL_G double mcomp() const { return f(LT_MCOMP); }
    SRL_G double com(int k) const { return f(LT_COM + k); }
    SRL_G double in(int k) const { return f(LT_IN + k); }
    SRL_G double F(int k) const { return f(LT_F + k); }
    SRL_G double t(int k) const { return f(LT_T + k); }
    SRL_G double ax(int k) const { return f(LT_AX + k); }
    SRL_G double jlo() const { return f(LT_JLO); }
    SRL_G double jhi() const { return f(LT_JHI); }
    SRL_G double damping() const { return f(LT_DAMP); }
    SRL_G double kp() const { return f(LT_KP); }
    SRL_G double bound() const { return f(LT_BOUND); }
    SRL_G double maxvel() const { return f(LT_MAXVEL); }
    SRL_G double q0() const { return f(LT_Q0); }
    SRL_G double tsel() const { return f(LT_TSEL); }
    SRL_G double sph(int k) const { return f(LT_SPH + k); }
    SRL_G double smu() const { return f(LT_SMU); }
    SRL_G uint32_t anc() const { return (uint32_t)f(LT_ANC); }
    SRL_G uint32_t desc() const { return (uint32_t)f(LT_DESC); }
    SRL_G int src(int k) const { return (int)f(LT_SRC + k); }
    SRL_G int slink() const { return (int)f(LT_SLINK); }
    SRL_G uint32_t sanc() const { return (uint32_t)f(LT_SANC); }
    SRL_G double eept(int k) const { return sc[LS_EEPT + k]; }
    SRL_G double grpt(int k) const { return sc[LS_GRPT + k]; }
    SRL_G double table_z() const { return sc[LS_TABLEZ]; }
    SRL_G double base_z() const { return sc[LS_BASEZ]; }
    SRL_G int ee_link() const { return (int)sc[LS_EELINK]; }
    SRL_G int grip_link() const { return (int)sc[LS_GRIPLINK]; }
    SRL_G int max_gen() const { return (int)sc[LS_MAXGEN]; }
    SRL_G bool friction() const { return sc[LS_FRICTION] != 0.0; }
    SRL_G int detail() const { return (int)sc[LS_DETAIL]; }
    SRL_G int ng() const { return (int)sc[LS_NG]; }
    SRL_G double contact_erp() const { return sc[LS_CERP]; }
    SRL_G double limit_erp() const { return sc[LS_LERP]; }
    SRL_G double linear_slop() const { return sc[LS_SLOP]; }
};
SRL_G TL lane_view(const double *tab) {
    // (the OFFSETS are laundered, not the pointer: an opaque pointer loses its LDS address space and every read becomes a flat
    //  load — vector-memory instructions that wait on vmcnt, i.e. on the previous step's output stores)
    int l = lane_id(), off = LT_COUNT * GL;
#if SRL_G_DEVICE
    asm volatile("" : "+v"(l), "+v"(off));
#endif
    TL L;
    L.l = l; L.p = tab + l; L.sc = tab + off;
    L.jnt = L.l < NJ; L.arm = L.l < NA; L.jm = L.jnt ? 1.0 : 0.0; L.am = L.arm ? 1.0 : 0.0;
    return L;
}

SRL_G void make_mask(uint32_t bits, double m[NJ]) {      // one of the two at a time: 24 registers instead of 48
#if SRL_G_DEVICE
    asm volatile("" : "+v"(bits));
#endif
#pragma unroll
    for (int k = 0; k < NJ; k++) m[k] = (bits >> k) & 1u ? 1.0 : 0.0;
}

// ------------------------------------------------------------------ kinematics
// world frame of every link: local joint transform per lane, then pointer jumping (4 levels cover the depth-10 finger tips)
SRL_G void tfk(const TL &L, GState &g) {
    const double s = g.sq, c = g.cq, v = 1.0 - c;         // lanes off the tree carry s = 0, c = 1: the identity
    double Rq[9], R[9], p[3];
    // Rodrigues, columns: Rq = c I + s [a]x + (1 - c) a a^T
    const double ax = L.ax(0), ay = L.ax(1), az = L.ax(2);
    Rq[0] = c + ax * ax * v;      Rq[1] = ay * ax * v + az * s; Rq[2] = az * ax * v - ay * s;
    Rq[3] = ax * ay * v - az * s; Rq[4] = c + ay * ay * v;      Rq[5] = az * ay * v + ax * s;
    Rq[6] = ax * az * v + ay * s; Rq[7] = ay * az * v - ax * s; Rq[8] = c + az * az * v;
#pragma unroll
    for (int j = 0; j < 3; j++)
#pragma unroll
        for (int k = 0; k < 3; k++) R[3 * j + k] = L.F(k) * Rq[3 * j] + L.F(3 + k) * Rq[3 * j + 1] + L.F(6 + k) * Rq[3 * j + 2];
    p[0] = L.t(0); p[1] = L.t(1); p[2] = L.t(2);
#pragma unroll
    for (int lvl = 0; lvl < 4; lvl++) {
        const int from = L.src(lvl);                   // parent^(2^lvl), or the identity lane 15 beyond the root
        double Ra[9], pa[3], Ro[9], po[3];
#pragma unroll
        for (int k = 0; k < 9; k++) Ra[k] = shfl(R[k], from);
#pragma unroll
        for (int k = 0; k < 3; k++) pa[k] = shfl(p[k], from);
        compose(Ra, pa, R, p, Ro, po);
#pragma unroll
        for (int k = 0; k < 9; k++) R[k] = Ro[k];
#pragma unroll
        for (int k = 0; k < 3; k++) p[k] = po[k];
    }
#pragma unroll
    for (int k = 0; k < 9; k++) g.R[k] = R[k];
#pragma unroll
    for (int k = 0; k < 3; k++) g.p[k] = p[k];
}
// world frame of link `link` (a per-lane index), fetched from its lane
SRL_G void link_frame(const GState &g, int link, double Rt[9], double pt[3]) {
#pragma unroll
    for (int k = 0; k < 9; k++) Rt[k] = shfl(g.R[k], link);
#pragma unroll
    for (int k = 0; k < 3; k++) pt[k] = shfl(g.p[k], link);
}
SRL_G void frame_point(const double R[9], const double p[3], const double local[3], double w[3]) {
#pragma unroll
    for (int k = 0; k < 3; k++) w[k] = p[k] + R[k] * local[0] + R[3 + k] * local[1] + R[6 + k] * local[2];
}
SRL_G void trefresh(const TL &L, GState &g, Env &e) {
    if (L.jnt) sincos(g.q, &g.sq, &g.cq); else { g.sq = 0.0; g.cq = 1.0; }
    tfk(L, g);
    double Rt[9], pt[3];
    link_frame(g, L.grip_link(), Rt, pt);
    const double grpt[3] = {L.grpt(0), L.grpt(1), L.grpt(2)};
    frame_point(Rt, pt, grpt, e.grip);
}

// ------------------------------------------------------------------ row-broadcast helpers for 12 joint lanes
template <int K> SRL_G void msum_step(double &acc, double x, const double m[NJ]) {
    fmac_bcast<K>(acc, x, m[K]);
    if constexpr (K + 1 < NJ) msum_step<K + 1>(acc, x, m);
}
SRL_G double msum(double x, const double m[NJ], double base = 0.0) { double acc = base; msum_step<0>(acc, x, m); return acc; }
// three masked sums at once: a serial chain of 12 dependent v_fmac_f64_dpp costs ~9 cycles per term (+ the DPP hazard nop); with
// three independent chains interleaved in one statement per source lane every instruction issues back to back
template <int K> SRL_G void msum3_step(double &a0, double &a1, double &a2, double x0, double x1, double x2, const double m[NJ]) {
#if SRL_G_DEVICE
    asm("s_nop 1\n\tv_fmac_f64_dpp %0, %3, %6 row_newbcast:%7 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %1, %4, %6 row_newbcast:%7 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %2, %5, %6 row_newbcast:%7 row_mask:0xf bank_mask:0xf"
        : "+v"(a0), "+v"(a1), "+v"(a2) : "v"(x0), "v"(x1), "v"(x2), "v"(m[K]), "n"(K));
#else
    a0 = fma(grp::host_exchange(x0, K), m[K], a0); a1 = fma(grp::host_exchange(x1, K), m[K], a1); a2 = fma(grp::host_exchange(x2, K), m[K], a2);
#endif
    if constexpr (K + 1 < NJ) msum3_step<K + 1>(a0, a1, a2, x0, x1, x2, m);
}
SRL_G void msum3(const double x[3], const double m[NJ], double out[3], double base2 = 0.0) {
    double a0 = 0.0, a1 = 0.0, a2 = base2;
    msum3_step<0>(a0, a1, a2, x[0], x[1], x[2], m);
    out[0] = a0; out[1] = a1; out[2] = a2;
}
template <int K, int N> SRL_G void ball_step(double x, double *out) {
    out[K] = bcast<K>(x);
    if constexpr (K + 1 < N) ball_step<K + 1, N>(x, out);
}
// out[K] = sum_c a[c] * bcast_K(b[c]) for K = 0..11, accumulated column by column: twelve independent chains per statement
SRL_G void dot6_all12(const double a[6], const double b[6], double out[NJ]) {
#pragma unroll
    for (int k = 0; k < NJ; k++) out[k] = 0.0;
#pragma unroll
    for (int c = 0; c < 6; c++) {
#if SRL_G_DEVICE
#define SRL_F(K) "v_fmac_f64_dpp %" #K ", %12, %13 row_newbcast:" #K " row_mask:0xf bank_mask:0xf\n\t"
        asm("s_nop 1\n\t" SRL_F(0) SRL_F(1) SRL_F(2) SRL_F(3) SRL_F(4) SRL_F(5) SRL_F(6) SRL_F(7) SRL_F(8) SRL_F(9) SRL_F(10) SRL_F(11)
            : "+v"(out[0]), "+v"(out[1]), "+v"(out[2]), "+v"(out[3]), "+v"(out[4]), "+v"(out[5]), "+v"(out[6]), "+v"(out[7]), "+v"(out[8]),
              "+v"(out[9]), "+v"(out[10]), "+v"(out[11])
            : "v"(b[c]), "v"(a[c]));
#undef SRL_F
#else
        for (int k = 0; k < NJ; k++) out[k] = fma(grp::host_exchange(b[c], k), a[c], out[k]);
#endif
    }
}
template <int K, int N> SRL_G void dot6_step(const double a[6], const double b[6], double *out) {
    double acc = 0.0;
#pragma unroll
    for (int c = 0; c < 6; c++) fmac_bcast<K>(acc, b[c], a[c]);
    out[K] = acc;
    if constexpr (K + 1 < N) dot6_step<K + 1, N>(a, b, out);
}
// M[K] on lane j < K takes lane K's low[j] (the mirrored half of the mass matrix).  Ordered by j, then K: the targets M[K] of
// consecutive instructions are independent (ordered by K, every M[K] was a serial chain of K dependent DPP FMAs).
template <int J, int K> SRL_G void transpose_inner(const TL &L, const double low[NJ], double M[NJ]) {
    fmac_bcast<K>(M[K], low[J], L.e(J));
    if constexpr (K + 1 < NJ) transpose_inner<J, K + 1>(L, low, M);
}
template <int J> SRL_G void transpose_step(const TL &L, const double low[NJ], double M[NJ]) {
    transpose_inner<J, J + 1>(L, low, M);
    if constexpr (J + 2 < NJ) transpose_step<J + 1>(L, low, M);
}
template <int K, int N> SRL_G void rdot_step(double &acc, const double *row, double x) {
    fmac_bcast<K>(acc, x, row[K]);
    if constexpr (K + 1 < N) rdot_step<K + 1, N>(acc, row, x);
}
// In-place Gauss-Jordan on an N x N SPD matrix, row i on lane i (lanes >= N carry zero rows).  INV: A <- A^-1, else A x = b -> b.
template <int K, int N, bool INV> SRL_G void gj_step(const TL &L, double *A, double &b, double *det = nullptr) {
    const double piv = bcast<K>(A[K]);
    if (det) *det *= piv;                    // product of the pivots = determinant (the IK's conditioning flag)
    const double r = rcp(piv);
    const double g = -((A[K] - L.e(K)) * r);
    if constexpr (INV) {
        A[K] = L.e(K);
#if SRL_G_DEVICE
        if constexpr (N == NJ) {
            // the twelve column updates of a pivot are independent: one statement, one hazard nop
#define SRL_F(C) "v_fmac_f64_dpp %" #C ", %" #C ", %12 row_newbcast:%13 row_mask:0xf bank_mask:0xf\n\t"
            asm("s_nop 1\n\t" SRL_F(0) SRL_F(1) SRL_F(2) SRL_F(3) SRL_F(4) SRL_F(5) SRL_F(6) SRL_F(7) SRL_F(8) SRL_F(9) SRL_F(10) SRL_F(11)
                : "+v"(A[0]), "+v"(A[1]), "+v"(A[2]), "+v"(A[3]), "+v"(A[4]), "+v"(A[5]), "+v"(A[6]), "+v"(A[7]), "+v"(A[8]), "+v"(A[9]),
                  "+v"(A[10]), "+v"(A[11])
                : "v"(g), "n"(K));
#undef SRL_F
        } else
#endif
        {
#pragma unroll
            for (int c = 0; c < N; c++) fmac_bcast<K>(A[c], A[c], g);
        }
    } else {
#pragma unroll
        for (int c = K + 1; c < N; c++) fmac_bcast<K>(A[c], A[c], g);
        fmac_bcast<K>(b, b, g);
    }
    if constexpr (K + 1 < N) gj_step<K + 1, N, INV>(L, A, b, det);
}

// ------------------------------------------------------------------ PGS, bank A only (no limit / contact row in the wavefront)
struct TRows {
    double acc0, cs;
    double n[GL];              // scaled couplings of the own bank-A row to bank-A rows (n[own] = 0)
    double diag, lo, S, jb;
};
// One sweep: 12 motor rows; the button's three rows are decoupled from them here and ride on rows 0..2.
// A whole sweep is TWO asm statements (an asm statement takes at most 30 operands; the compiler pads every asm boundary with wait
// states): rows 0..5 with the button rows riding on 0..2, rows 6..11.  39 VALU instructions per sweep.
struct TEs { double e[NJ]; };          // e[j] = (lane == j), plus the button lanes on 0..2: who restarts its accumulator after row j
SRL_G void sweep_free(const TEs &E, const TRows &r, double &acc, double ep_first) {
#if SRL_G_DEVICE
    double t;
#define SRL_ROW(J, NJ_, EP) "v_add_f64 %1, %2, %0 clamp\n\tv_fma_f64 %0, -%" #EP ", %0, %0\n\ts_nop 0\n\tv_fmac_f64_dpp %0, %1, %" #NJ_ " row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n\t"
#define SRL_ROWB(J, NJ_) "v_fmac_f64_dpp %0, %1, %" #NJ_ " row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n\t"
    asm volatile(SRL_ROW(0, 3, 12) SRL_ROWB(12, 9) SRL_ROW(1, 4, 13) SRL_ROWB(13, 10) SRL_ROW(2, 5, 14) SRL_ROWB(14, 11)
                 SRL_ROW(3, 6, 15) SRL_ROW(4, 7, 16) SRL_ROW(5, 8, 17)
                 : "+v"(acc), "=&v"(t)
                 : "v"(r.cs), "v"(r.n[0]), "v"(r.n[1]), "v"(r.n[2]), "v"(r.n[3]), "v"(r.n[4]), "v"(r.n[5]),          // %2 .. %8
                   "v"(r.n[kBM]), "v"(r.n[kBLo]), "v"(r.n[kBHi]),                                                     // %9 .. %11
                   "v"(ep_first), "v"(E.e[0]), "v"(E.e[1]), "v"(E.e[2]), "v"(E.e[3]), "v"(E.e[4]));                   // %12 .. %17
    asm volatile(SRL_ROW(6, 3, 9) SRL_ROW(7, 4, 10) SRL_ROW(8, 5, 11) SRL_ROW(9, 6, 12) SRL_ROW(10, 7, 13) SRL_ROW(11, 8, 14)
                 : "+v"(acc), "=&v"(t)
                 : "v"(r.cs), "v"(r.n[6]), "v"(r.n[7]), "v"(r.n[8]), "v"(r.n[9]), "v"(r.n[10]), "v"(r.n[11]),        // %2 .. %8
                   "v"(E.e[5]), "v"(E.e[6]), "v"(E.e[7]), "v"(E.e[8]), "v"(E.e[9]), "v"(E.e[10]));                     // %9 .. %14
#undef SRL_ROW
#undef SRL_ROWB
#else
    pgs_row2<0, kBM>(acc, r.cs, r.n[0], r.n[kBM], ep_first);
    pgs_row2<1, kBLo>(acc, r.cs, r.n[1], r.n[kBLo], E.e[0]);
    pgs_row2<2, kBHi>(acc, r.cs, r.n[2], r.n[kBHi], E.e[1]);
    pgs_row<3>(acc, r.cs, r.n[3], E.e[2]);    pgs_row<4>(acc, r.cs, r.n[4], E.e[3]);  pgs_row<5>(acc, r.cs, r.n[5], E.e[4]);
    pgs_row<6>(acc, r.cs, r.n[6], E.e[5]);    pgs_row<7>(acc, r.cs, r.n[7], E.e[6]);  pgs_row<8>(acc, r.cs, r.n[8], E.e[7]);
    pgs_row<9>(acc, r.cs, r.n[9], E.e[8]);    pgs_row<10>(acc, r.cs, r.n[10], E.e[9]); pgs_row<11>(acc, r.cs, r.n[11], E.e[10]);
#endif
}
// Kuka2Button: the second button's three rows live on the same lanes (kBM, kBLo, kBHi) with their own accumulator.  Without a
// contact they couple to nothing but each other: an independent three-row chain per sweep.
struct TRows2 { double cs, n[3], lo, S, jb; };
SRL_G void sweep_free2(const TRows2 &r2, double &acc2, double e_bm, double e_blo, double e_bhi_prev) {
    pgs_row<kBM>(acc2, r2.cs, r2.n[0], e_bhi_prev);
    pgs_row<kBLo>(acc2, r2.cs, r2.n[1], e_bm);
    pgs_row<kBHi>(acc2, r2.cs, r2.n[2], e_blo);
}
template <int NB = 1>
SRL_G double sweeps_free(const TRows &r, const TRows2 *r2 = nullptr, double *u2_out = nullptr) {
    double acc = r.acc0, u = 0.0, t;
    int l = lane_id();
#if SRL_G_DEVICE
    asm volatile("" : "+v"(l));          // the e's are rebuilt here: nothing of them is live outside the solver
#endif
    TEs E;
#pragma unroll
    for (int j = 0; j < NJ; j++) E.e[j] = l == j ? 1.0 : 0.0;
    E.e[0] += l == kBM ? 1.0 : 0.0; E.e[1] += l == kBLo ? 1.0 : 0.0; E.e[2] += l == kBHi ? 1.0 : 0.0;
    const double e0 = E.e[0], e1 = E.e[1], e2 = E.e[2];
    double acc2 = 0.0, u2 = 0.0;
    const double c_bm = l == kBM ? 1.0 : 0.0, c_blo = l == kBLo ? 1.0 : 0.0, c_bhi = l == kBHi ? 1.0 : 0.0;
    sweep_free(E, r, acc, 0.0);
    if constexpr (NB == 2) sweep_free2(*r2, acc2, c_bm, c_blo, 0.0);
    for (int it = 1; it < kSolverIters - 1; it++) {
        sweep_free(E, r, acc, E.e[11]);
        if constexpr (NB == 2) sweep_free2(*r2, acc2, c_bm, c_blo, c_bhi);
    }
    if constexpr (NB == 2) {
        t = pgs_row<kBM>(acc2, r2->cs, r2->n[0], c_bhi);   u2 = fma(c_bm, t, u2);
        t = pgs_row<kBLo>(acc2, r2->cs, r2->n[1], c_bm);   u2 = fma(c_blo, t, u2);
        t = pgs_row<kBHi>(acc2, r2->cs, r2->n[2], c_blo);  u2 = fma(c_bhi, t, u2);
        *u2_out = u2;
    }
    t = pgs_row2<0, kBM>(acc, r.cs, r.n[0], r.n[kBM], E.e[11]);  u = fma(e0, t, u);
    t = pgs_row2<1, kBLo>(acc, r.cs, r.n[1], r.n[kBLo], e0);      u = fma(e1, t, u);
    t = pgs_row2<2, kBHi>(acc, r.cs, r.n[2], r.n[kBHi], e1);      u = fma(e2, t, u);
    t = pgs_row<3>(acc, r.cs, r.n[3], e2);         u = fma(E.e[3], t, u);
    t = pgs_row<4>(acc, r.cs, r.n[4], E.e[3]);     u = fma(E.e[4], t, u);
    t = pgs_row<5>(acc, r.cs, r.n[5], E.e[4]);     u = fma(E.e[5], t, u);
    t = pgs_row<6>(acc, r.cs, r.n[6], E.e[5]);     u = fma(E.e[6], t, u);
    t = pgs_row<7>(acc, r.cs, r.n[7], E.e[6]);     u = fma(E.e[7], t, u);
    t = pgs_row<8>(acc, r.cs, r.n[8], E.e[7]);     u = fma(E.e[8], t, u);
    t = pgs_row<9>(acc, r.cs, r.n[9], E.e[8]);     u = fma(E.e[9], t, u);
    t = pgs_row<10>(acc, r.cs, r.n[10], E.e[9]);   u = fma(E.e[10], t, u);
    t = pgs_row<11>(acc, r.cs, r.n[11], E.e[10]);  u = fma(E.e[11], t, u);
    return u;
}

// ------------------------------------------------------------------ PGS, general path: bank A + bank B
struct BRow { double cs, lo, hi, mu, lam, jb, inv_diag; int normal; bool on, fric;      // own bank-B row (slot == lane)
              int bsel; double nBC[3];      // Kuka2Button: the glider the row acts on; its scaled couplings to the second button's rows
              int obj; double Jo[3], jo2m; };   // KukaRandButton: the free body the row also acts on (-1: none), its Jacobian there, Jo . Jo / m
// bank-A row J: u = clamp01(cs + accA) on lane J, broadcast to both accumulators of every lane.  (The general path resets the
// own accumulator explicitly instead of in the shadow of the next row: bank-B rows interleave with bank A.)
template <int J> SRL_G void gen_rowA(const TL &L, const TRows &r, const double *sc, double &accA, double &accB, double &uA) {
    const double t = clamp01(r.cs + accA);
    const double tb = bcast<J>(t);
    if (L.l == J) { accA = 0.0; uA = t; }
    accA = fma(r.n[J], tb, accA);
    accB = fma(sc[SC_NBA + J * GL + L.l], tb, accB);
}
// Kuka2Button, row J (kBM / kBLo / kBHi) of the second button: same lanes, own accumulator accC; couples to the bank-B rows that act on glider 2
template <int J> SRL_G void gen_rowC(const TL &L, const TRows2 &r2, const BRow &b, double &accC, double &accB, double &uC) {
    const double t = clamp01(r2.cs + accC);
    const double tb = bcast<J>(t);
    if (L.l == J) { accC = 0.0; uC = t; }
    accC = fma(r2.n[J - kBM], tb, accC);
    accB = fma(b.nBC[J - kBM], tb, accB);
}
// bank-B slot s (a wave-uniform loop index): lambda = clamp(cs + accB, lo, hi) on lane s.  A friction row's bounds are +-mu times
// the CURRENT impulse of its normal row, and the row keeps its value while that is not positive (it still hands the value round:
// every row contributes exactly once per sweep to the others' accumulators).  part: this env's slot s belongs to the phase being
// swept (uniform over the env's 16 lanes) — otherwise nothing of this env changes.
template <int NB = 1>
SRL_G void gen_rowB(const TL &L, const double *sc, int s, BRow &b, double &accA, double &accB, bool part, double *accC = nullptr, double nCB_s = 0.0) {
    double lo = b.lo, hi = b.hi;
    const double tot = shfl(b.lam, b.normal);               // friction rows: the normal row's impulse
    const bool skip = b.fric && !(tot > 0.0);
    if (b.fric) { lo = -b.mu * tot; hi = b.mu * tot; }
    double t = b.cs + accB;
    t = t < lo ? lo : (t > hi ? hi : t);
    if (skip) t = b.lam;
    const double tb = shfl(t, s);
    const double nA = fma(sc[SC_NAB + s * GL + L.l], tb, accA);
    const double nB = fma(sc[SC_NBB + s * GL + L.l], tb, L.l == s ? 0.0 : accB);
    if (part) { accA = nA; accB = nB; if (L.l == s) b.lam = t; }
    if constexpr (NB == 2) { if (part) *accC = fma(nCB_s, tb, *accC); }
}

// Contact steps without a joint-limit row in the wavefront (every contact case a random agent produces: two finger tips with two
// spheres each on the cap, plus the base): all couplings in registers and every broadcast a DPP row_newbcast — no LDS access and
// no ds_bpermute inside the 150 sweeps (the general loop below pays ~100 cycles of LDS latency per row: measured 300+ us per
// step with four contacts).  Slots: contact normals g = 0..kNGen-1 on lanes g, their friction rows on lanes kNGen + g; slots a
// wavefront does not use (g >= ngen_w, wave-uniform) are skipped, slots an ENV does not use have zero coefficients.
template <int J, bool LAST> SRL_G void cn_rowA(const TRows &r, double nBA_J, double eJ, double &accA, double &accB, double &uA) {
#if SRL_G_DEVICE
    double t;
    // t = clamp01(cs + accA); the own accumulator restarts; both banks take n * bcast_J(t)  (one statement: nothing is scheduled into it)
    asm volatile("v_add_f64 %2, %3, %0 clamp\n\tv_fma_f64 %0, -%6, %0, %0\n\ts_nop 0\n\t"
                 "v_fmac_f64_dpp %0, %2, %4 row_newbcast:%7 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %1, %2, %5 row_newbcast:%7 row_mask:0xf bank_mask:0xf"
                 : "+v"(accA), "+v"(accB), "=&v"(t) : "v"(r.cs), "v"(r.n[J]), "v"(nBA_J), "v"(eJ), "n"(J));
#else
    const double t = clamp01(r.cs + accA);
    accA = fma(-eJ, accA, accA);                    // the own accumulator restarts
    accA = fma(grp::host_exchange(t, J), r.n[J], accA);
    accB = fma(grp::host_exchange(t, J), nBA_J, accB);
#endif
    if (LAST) uA = fma(eJ, t - uA, uA);             // lane J keeps its value (last sweep only)
}
// contact-normal slot G on lane G: lambda = clamp(cs + accB, 0, hi)
template <int J, bool LAST> SRL_G void cn_rowC(const TRows2 &r2, double nBC_J, double eJ, double &accC, double &accB, double &uC) {
    const double t = clamp01(r2.cs + accC);
    accC = fma(-eJ, accC, accC);
    fmac_bcast<J>(accC, t, r2.n[J - kBM]);
    fmac_bcast<J>(accB, t, nBC_J);
    if (LAST) uC = fma(eJ, t - uC, uC);
}
// Bank-B rows of the contact path.  Only the OWNING lane's value of a row is ever consumed (through a row broadcast), so a row's
// impulse is simply the register `t` of its last update on every lane — no masked bookkeeping: tN[G] / tF[G] are the impulses of
// normal slot G (lane G) and of its friction row (lane kNGen + G).  One asm statement per row for the DPP part (one hazard nop).
template <int S, int NB> SRL_G void cn_spread(double t, double nAB, double nBB, double nCB, double &accA, double &accB, double *accC) {
#if SRL_G_DEVICE
    if constexpr (NB == 2)
        asm("s_nop 1\n\tv_fmac_f64_dpp %0, %3, %4 row_newbcast:%7 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %1, %3, %5 row_newbcast:%7 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %2, %3, %6 row_newbcast:%7 row_mask:0xf bank_mask:0xf"
            : "+v"(accA), "+v"(accB), "+v"(*accC) : "v"(t), "v"(nAB), "v"(nBB), "v"(nCB), "n"(S));
    else
        asm("s_nop 1\n\tv_fmac_f64_dpp %0, %2, %3 row_newbcast:%5 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %1, %2, %4 row_newbcast:%5 row_mask:0xf bank_mask:0xf"
            : "+v"(accA), "+v"(accB) : "v"(t), "v"(nAB), "v"(nBB), "n"(S));
#else
    const double tb = grp::host_exchange(t, S);
    accA = fma(tb, nAB, accA); accB = fma(tb, nBB, accB);
    if constexpr (NB == 2) *accC = fma(tb, nCB, *accC);
#endif
}
// contact-normal slot G on lane G: lambda = clamp(cs + accB, 0, hi)   (an unused slot has cs = 0, zero couplings and hi = 0: 0)
template <int G, int NB = 1> SRL_G void cn_rowN(const BRow &b, double &tN, double nAB, double nBB, double eS, double &accA, double &accB, double *accC = nullptr, double nCB = 0.0) {
    const double t = fmin(fmax(b.cs + accB, 0.0), b.hi);
    accB = fma(-eS, accB, accB);                    // the own accumulator restarts
    tN = t;
    cn_spread<G, NB>(t, nAB, nBB, nCB, accA, accB, accC);
}
// friction slot kNGen + G: bounds +-mu * (current impulse of normal slot G); the row keeps its value while that is not positive
// (mu > 0 on a lane that owns an active friction row, 0 elsewhere: hi > 0 says both)
template <int G, int NB = 1> SRL_G void cn_rowF(const BRow &b, double tN, double &tF, double nAB, double nBB, double eS, double &accA, double &accB, double *accC = nullptr, double nCB = 0.0) {
    double hi = 0.0;
    fmac_bcast<G>(hi, tN, b.mu);                    // mu * (impulse of normal slot G), one DPP instruction (0 + a b rounds like a b)
    double t = fmin(fmax(b.cs + accB, -hi), hi);
    t = hi > 0.0 ? t : tF;
    accB = fma(-eS, accB, accB);
    tF = t;
    cn_spread<kNGen + G, NB>(t, nAB, nBB, nCB, accA, accB, accC);
}
template <int G, int NB = 1> SRL_G void cn_normals(const BRow &b, double *tN, const double *nAB, const double *nBB, const double *eB, double &accA, double &accB, int ngen_w,
                                                   double *accC = nullptr, const double *nCB = nullptr) {
    if (G < ngen_w) cn_rowN<G, NB>(b, tN[G], nAB[G], nBB[G], eB[G], accA, accB, accC, NB == 2 ? nCB[G] : 0.0);
    if constexpr (G + 1 < kNGen) cn_normals<G + 1, NB>(b, tN, nAB, nBB, eB, accA, accB, ngen_w, accC, nCB);
}
template <int G, int NB = 1> SRL_G void cn_frictions(const BRow &b, const double *tN, double *tF, const double *nAB, const double *nBB, const double *eB, double &accA, double &accB, int ngen_w,
                                                     double *accC = nullptr, const double *nCB = nullptr) {
    if (G < ngen_w) cn_rowF<G, NB>(b, tN[G], tF[G], nAB[kNGen + G], nBB[kNGen + G], eB[kNGen + G], accA, accB, accC, NB == 2 ? nCB[kNGen + G] : 0.0);
    if constexpr (G + 1 < kNGen) cn_frictions<G + 1, NB>(b, tN, tF, nAB, nBB, eB, accA, accB, ngen_w, accC, nCB);
}
// the own row's impulse after the last sweep: lane l < kNGen owns normal slot l, lane kNGen + g the friction row of slot g
SRL_G double cn_own_lambda(const double *tN, const double *tF, int l) {
    double lam = 0.0;
#pragma unroll
    for (int g = 0; g < kNGen; g++) { lam = l == g ? tN[g] : lam; lam = l == kNGen + g ? tF[g] : lam; }
    return lam;
}
// Round 4, one-button contact sweeps: (1) the button's three rows ride on motor rows 0..2 as on the free path — a motor row and a
// button row never couple directly (only through bank-B rows, which come after both), so updating the pair at once changes no
// value and takes three rows out of the sequential chain; (2) contact-normal rows live in u = lambda / 2^33 so that their projection
// [0, 1e10] becomes the hardware clamp of the add (upper bound 2^33 = 8.6e9 instead of 1e10: impulses are ~1e-2): two dependent
// operations per normal row instead of four.  The scaling is a power of two: exact.
constexpr double kNormalScale = 8589934592.0;          // 2^33
template <int J, int J2, bool LAST> SRL_G void cn_rowA2(const TRows &r, double nBA_J, double nBA_J2, double eJJ2, double &accA, double &accB, double &uA) {
#if SRL_G_DEVICE
    double t;
    asm volatile("v_add_f64 %2, %3, %0 clamp\n\tv_fma_f64 %0, -%8, %0, %0\n\ts_nop 0\n\t"
                 "v_fmac_f64_dpp %0, %2, %4 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %2, %5 row_newbcast:%10 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %1, %2, %6 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %1, %2, %7 row_newbcast:%10 row_mask:0xf bank_mask:0xf"
                 : "+v"(accA), "+v"(accB), "=&v"(t)
                 : "v"(r.cs), "v"(r.n[J]), "v"(r.n[J2]), "v"(nBA_J), "v"(nBA_J2), "v"(eJJ2), "n"(J), "n"(J2));
#else
    const double t = clamp01(r.cs + accA);
    accA = fma(-eJJ2, accA, accA);
    const double ta = grp::host_exchange(t, J), tb = grp::host_exchange(t, J2);
    accA = fma(ta, r.n[J], accA); accA = fma(tb, r.n[J2], accA);
    accB = fma(ta, nBA_J, accB); accB = fma(tb, nBA_J2, accB);
#endif
    if (LAST) uA = fma(eJJ2, t - uA, uA);
}
// scaled contact-normal slot G on lane G: u = clamp01(cs_u + accB)
template <int G> SRL_G void cn_rowNs(double csU, double &tN, double nAB, double nBB, double eS, double &accA, double &accB) {
#if SRL_G_DEVICE
    double t;
    asm volatile("v_add_f64 %2, %3, %1 clamp\n\tv_fma_f64 %1, -%6, %1, %1\n\ts_nop 0\n\t"
                 "v_fmac_f64_dpp %0, %2, %4 row_newbcast:%7 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %1, %2, %5 row_newbcast:%7 row_mask:0xf bank_mask:0xf"
                 : "+v"(accA), "+v"(accB), "=&v"(t) : "v"(csU), "v"(nAB), "v"(nBB), "v"(eS), "n"(G));
#else
    const double t = clamp01(csU + accB);
    accB = fma(-eS, accB, accB);
    const double tb = grp::host_exchange(t, G);
    accA = fma(tb, nAB, accA); accB = fma(tb, nBB, accB);
#endif
    tN = t;
}
// The one-button sweep loop with the number of bank-B slots in use as a COMPILE-TIME constant: with a run-time count every slot
// costs a check (the compiler round-trips the condition through a VGPR: four instructions per slot, 48 of the loop's 261), and
// contact steps — four to six per cent of all steps, 160 k cycles each, which is also what a single-step launch lasts — are
// this loop 150 times.
template <int G, int NG> SRL_G void cn_normals_sn(double csU, double *tN, const double *nAB, const double *nBB, const double *eB, double &accA, double &accB) {
    if constexpr (G < NG) {
        cn_rowNs<G>(csU, tN[G], nAB[G], nBB[G], eB[G], accA, accB);
        cn_normals_sn<G + 1, NG>(csU, tN, nAB, nBB, eB, accA, accB);
    }
}
template <int G, int NG> SRL_G void cn_frictions_n(const BRow &b, const double *tN, double *tF, const double *nAB, const double *nBB, const double *eB, double &accA, double &accB) {
    if constexpr (G < NG) {
        cn_rowF<G>(b, tN[G], tF[G], nAB[kNGen + G], nBB[kNGen + G], eB[kNGen + G], accA, accB);
        cn_frictions_n<G + 1, NG>(b, tN, tF, nAB, nBB, eB, accA, accB);
    }
}
// The bank-A rows of a one-button contact sweep (every sweep but the last) as three hand-scheduled statements.  A DPP operand
// written by the previous VALU instruction needs two wait states; a row written on its own fills them with the restart FMA and
// an s_nop and is followed by the compiler's statement padding: six issue slots.  Here the row's second fmac (its coupling into
// the bank-B accumulator, which nothing reads before the bank-B phase) is DEFERRED into the next row's wait states — rows
// alternate between two value registers — so a row is four slots: add, restart, previous row's accB fmac, own accA fmac
// (profiles/probes/pgs_row_timing.hip: 11.4 -> 9.2 ns per row; the dependent chain add -> fmac alone is 7.9).  Same operations
// on the same accumulators in the same order: bit-identical to cn_rowA / cn_rowA2.
SRL_G void cn_phaseA(const TRows &r, const double *nBA, const double *eA, double e0, double e1, double e2, double &accA, double &accB) {
#if SRL_G_DEVICE
    double t0, t1;
#define SRL_DPP(J) " row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n\t"
    asm volatile("v_add_f64 %[t0], %[cs], %[a] clamp\n\tv_fma_f64 %[a], -%[e0], %[a], %[a]\n\ts_nop 0\n\t"
                 "v_fmac_f64_dpp %[a], %[t0], %[n0]" SRL_DPP(0) "v_fmac_f64_dpp %[a], %[t0], %[n12]" SRL_DPP(12)
                 "v_add_f64 %[t1], %[cs], %[a] clamp\n\tv_fma_f64 %[a], -%[e1], %[a], %[a]\n\t"
                 "v_fmac_f64_dpp %[b], %[t0], %[m0]" SRL_DPP(0) "v_fmac_f64_dpp %[b], %[t0], %[m12]" SRL_DPP(12)
                 "v_fmac_f64_dpp %[a], %[t1], %[n1]" SRL_DPP(1) "v_fmac_f64_dpp %[a], %[t1], %[n13]" SRL_DPP(13)
                 "v_add_f64 %[t0], %[cs], %[a] clamp\n\tv_fma_f64 %[a], -%[e2], %[a], %[a]\n\t"
                 "v_fmac_f64_dpp %[b], %[t1], %[m1]" SRL_DPP(1) "v_fmac_f64_dpp %[b], %[t1], %[m13]" SRL_DPP(13)
                 "v_fmac_f64_dpp %[a], %[t0], %[n2]" SRL_DPP(2) "v_fmac_f64_dpp %[a], %[t0], %[n14]" SRL_DPP(14)
                 : [a] "+v"(accA), [b] "+v"(accB), [t0] "=&v"(t0), [t1] "=&v"(t1)
                 : [cs] "v"(r.cs), [e0] "v"(e0), [e1] "v"(e1), [e2] "v"(e2), [n0] "v"(r.n[0]), [n12] "v"(r.n[kBM]), [n1] "v"(r.n[1]), [n13] "v"(r.n[kBLo]),
                   [n2] "v"(r.n[2]), [n14] "v"(r.n[kBHi]), [m0] "v"(nBA[0]), [m12] "v"(nBA[kBM]), [m1] "v"(nBA[1]), [m13] "v"(nBA[kBLo]));
#define SRL_A1(T, TP, J, JP) "v_add_f64 %[" #T "], %[cs], %[a] clamp\n\tv_fma_f64 %[a], -%[e" #J "], %[a], %[a]\n\t"                 \
                             "v_fmac_f64_dpp %[b], %[" #TP "], %[m" #JP "]" SRL_DPP(JP) "v_fmac_f64_dpp %[a], %[" #T "], %[n" #J "]" SRL_DPP(J)
    asm volatile("v_add_f64 %[t1], %[cs], %[a] clamp\n\tv_fma_f64 %[a], -%[e3], %[a], %[a]\n\t"
                 "v_fmac_f64_dpp %[b], %[t0], %[m2]" SRL_DPP(2) "v_fmac_f64_dpp %[b], %[t0], %[m14]" SRL_DPP(14)
                 "v_fmac_f64_dpp %[a], %[t1], %[n3]" SRL_DPP(3)
                 SRL_A1(t0, t1, 4, 3) SRL_A1(t1, t0, 5, 4) SRL_A1(t0, t1, 6, 5) SRL_A1(t1, t0, 7, 6)
                 : [a] "+v"(accA), [b] "+v"(accB), [t0] "+v"(t0), [t1] "=&v"(t1)
                 : [cs] "v"(r.cs), [e3] "v"(eA[3]), [e4] "v"(eA[4]), [e5] "v"(eA[5]), [e6] "v"(eA[6]), [e7] "v"(eA[7]),
                   [n3] "v"(r.n[3]), [n4] "v"(r.n[4]), [n5] "v"(r.n[5]), [n6] "v"(r.n[6]), [n7] "v"(r.n[7]),
                   [m2] "v"(nBA[2]), [m14] "v"(nBA[kBHi]), [m3] "v"(nBA[3]), [m4] "v"(nBA[4]), [m5] "v"(nBA[5]), [m6] "v"(nBA[6]));
    asm volatile(SRL_A1(t0, t1, 8, 7) SRL_A1(t1, t0, 9, 8) SRL_A1(t0, t1, 10, 9) SRL_A1(t1, t0, 11, 10)
                 "v_fmac_f64_dpp %[b], %[t1], %[m11]" SRL_DPP(11)
                 : [a] "+v"(accA), [b] "+v"(accB), [t0] "=&v"(t0), [t1] "+v"(t1)
                 : [cs] "v"(r.cs), [e8] "v"(eA[8]), [e9] "v"(eA[9]), [e10] "v"(eA[10]), [e11] "v"(eA[11]),
                   [n8] "v"(r.n[8]), [n9] "v"(r.n[9]), [n10] "v"(r.n[10]), [n11] "v"(r.n[11]),
                   [m7] "v"(nBA[7]), [m8] "v"(nBA[8]), [m9] "v"(nBA[9]), [m10] "v"(nBA[10]), [m11] "v"(nBA[11]));
#undef SRL_A1
#undef SRL_DPP
#else
    double uA = 0.0;
    cn_rowA2<0, kBM, false>(r, nBA[0], nBA[kBM], e0, accA, accB, uA);   cn_rowA2<1, kBLo, false>(r, nBA[1], nBA[kBLo], e1, accA, accB, uA);
    cn_rowA2<2, kBHi, false>(r, nBA[2], nBA[kBHi], e2, accA, accB, uA);
    cn_rowA<3, false>(r, nBA[3], eA[3], accA, accB, uA);   cn_rowA<4, false>(r, nBA[4], eA[4], accA, accB, uA);   cn_rowA<5, false>(r, nBA[5], eA[5], accA, accB, uA);
    cn_rowA<6, false>(r, nBA[6], eA[6], accA, accB, uA);   cn_rowA<7, false>(r, nBA[7], eA[7], accA, accB, uA);   cn_rowA<8, false>(r, nBA[8], eA[8], accA, accB, uA);
    cn_rowA<9, false>(r, nBA[9], eA[9], accA, accB, uA);   cn_rowA<10, false>(r, nBA[10], eA[10], accA, accB, uA); cn_rowA<11, false>(r, nBA[11], eA[11], accA, accB, uA);
#endif
}
template <int NG> SRL_G double cn_sweeps(const TRows &r, const BRow &bb, const double *nBA, const double *eA, double e0, double e1, double e2, const double *nAB,
                                         const double *nBB, const double *eB, double &accA, double &accB, double &uA, int l) {
    double tN[kNGen], tF[kNGen];                        // (local to the instantiation: handed in by pointer they ended up in scratch in some kernels)
#pragma unroll
    for (int g = 0; g < kNGen; g++) { tN[g] = 0.0; tF[g] = 0.0; }
#define SRL_CN_SWEEP(LAST)                                                                                                                     \
    cn_rowA2<0, kBM, LAST>(r, nBA[0], nBA[kBM], e0, accA, accB, uA);   cn_rowA2<1, kBLo, LAST>(r, nBA[1], nBA[kBLo], e1, accA, accB, uA);      \
    cn_rowA2<2, kBHi, LAST>(r, nBA[2], nBA[kBHi], e2, accA, accB, uA);                                                                         \
    cn_rowA<3, LAST>(r, nBA[3], eA[3], accA, accB, uA);   cn_rowA<4, LAST>(r, nBA[4], eA[4], accA, accB, uA);   cn_rowA<5, LAST>(r, nBA[5], eA[5], accA, accB, uA);   \
    cn_rowA<6, LAST>(r, nBA[6], eA[6], accA, accB, uA);   cn_rowA<7, LAST>(r, nBA[7], eA[7], accA, accB, uA);   cn_rowA<8, LAST>(r, nBA[8], eA[8], accA, accB, uA);   \
    cn_rowA<9, LAST>(r, nBA[9], eA[9], accA, accB, uA);   cn_rowA<10, LAST>(r, nBA[10], eA[10], accA, accB, uA); cn_rowA<11, LAST>(r, nBA[11], eA[11], accA, accB, uA); \
    cn_normals_sn<0, NG>(bb.cs, tN, nAB, nBB, eB, accA, accB);                                                                                 \
    cn_frictions_n<0, NG>(bb, tN, tF, nAB, nBB, eB, accA, accB);
    for (int it = 0; it < kSolverIters - 1; it++) {
        cn_phaseA(r, nBA, eA, e0, e1, e2, accA, accB);
        cn_normals_sn<0, NG>(bb.cs, tN, nAB, nBB, eB, accA, accB);
        cn_frictions_n<0, NG>(bb, tN, tF, nAB, nBB, eB, accA, accB);
    }
    { SRL_CN_SWEEP(true) }                              // the last sweep keeps every row's own value (uA): row by row
#undef SRL_CN_SWEEP
    return cn_own_lambda(tN, tF, l);
}
// Kuka2Button: the same sweeps with the second button's rows in Bullet's order (motors, both button motors, both pairs of button
// stops, normals, frictions).  nCB: the second button's rows' couplings to the bank-B slots (per lane, zero off the button lanes).
SRL_G double sweeps_contacts2(const TRows &r, const TRows2 &r2, BRow &b, const double *sc, double accA, int ngen_w, const double *nCB, double *u2_out) {
    int l = lane_id();
#if SRL_G_DEVICE
    asm volatile("" : "+v"(l));
#endif
    double nBA[kNArows], eA[kNArows], nAB[kNB], nBB[kNB], eB[kNB];
#pragma unroll
    for (int j = 0; j < kNArows; j++) { nBA[j] = sc[SC_NBA + j * GL + l]; eA[j] = l == j ? 1.0 : 0.0; }
#pragma unroll
    for (int s = 0; s < kNB; s++) { nAB[s] = sc[SC_NAB + s * GL + l]; nBB[s] = sc[SC_NBB + s * GL + l]; eB[s] = l == s ? 1.0 : 0.0; }
    BRow bb = b;
    if (!(bb.on && !bb.fric)) bb.hi = 0.0;
    if (!bb.on) bb.cs = 0.0;
    double accB = 0.0, uA = 0.0, accC = 0.0, uC = 0.0, tN[kNGen], tF[kNGen];
#pragma unroll
    for (int g = 0; g < kNGen; g++) { tN[g] = 0.0; tF[g] = 0.0; }
#define SRL_CN_SWEEP(LAST)                                                                                                                     \
    cn_rowA<0, LAST>(r, nBA[0], eA[0], accA, accB, uA);   cn_rowA<1, LAST>(r, nBA[1], eA[1], accA, accB, uA);   cn_rowA<2, LAST>(r, nBA[2], eA[2], accA, accB, uA);   \
    cn_rowA<3, LAST>(r, nBA[3], eA[3], accA, accB, uA);   cn_rowA<4, LAST>(r, nBA[4], eA[4], accA, accB, uA);   cn_rowA<5, LAST>(r, nBA[5], eA[5], accA, accB, uA);   \
    cn_rowA<6, LAST>(r, nBA[6], eA[6], accA, accB, uA);   cn_rowA<7, LAST>(r, nBA[7], eA[7], accA, accB, uA);   cn_rowA<8, LAST>(r, nBA[8], eA[8], accA, accB, uA);   \
    cn_rowA<9, LAST>(r, nBA[9], eA[9], accA, accB, uA);   cn_rowA<10, LAST>(r, nBA[10], eA[10], accA, accB, uA); cn_rowA<11, LAST>(r, nBA[11], eA[11], accA, accB, uA); \
    cn_rowA<kBM, LAST>(r, nBA[kBM], eA[kBM], accA, accB, uA); cn_rowC<kBM, LAST>(r2, bb.nBC[0], eA[kBM], accC, accB, uC);                      \
    cn_rowA<kBLo, LAST>(r, nBA[kBLo], eA[kBLo], accA, accB, uA); cn_rowA<kBHi, LAST>(r, nBA[kBHi], eA[kBHi], accA, accB, uA);                  \
    cn_rowC<kBLo, LAST>(r2, bb.nBC[1], eA[kBLo], accC, accB, uC); cn_rowC<kBHi, LAST>(r2, bb.nBC[2], eA[kBHi], accC, accB, uC);                \
    cn_normals<0, 2>(bb, tN, nAB, nBB, eB, accA, accB, ngen_w, &accC, nCB);                                                                    \
    cn_frictions<0, 2>(bb, tN, tF, nAB, nBB, eB, accA, accB, ngen_w, &accC, nCB);
    for (int it = 0; it < kSolverIters - 1; it++) { SRL_CN_SWEEP(false) }
    { SRL_CN_SWEEP(true) }
#undef SRL_CN_SWEEP
    b.lam = cn_own_lambda(tN, tF, l);
    *u2_out = uC;
    return uA;
}
SRL_G double sweeps_contacts(const TRows &r, BRow &b, const double *sc, double accA, int ngen_w) {
    int l = lane_id();
#if SRL_G_DEVICE
    asm volatile("" : "+v"(l));
#endif
    double nBA[kNArows], eA[kNArows], nAB[kNB], nBB[kNB], eB[kNB];
#pragma unroll
    for (int j = 0; j < kNArows; j++) { nBA[j] = sc[SC_NBA + j * GL + l]; eA[j] = l == j ? 1.0 : 0.0; }
#pragma unroll
    for (int s = 0; s < kNB; s++) { nAB[s] = sc[SC_NAB + s * GL + l]; nBB[s] = sc[SC_NBB + s * GL + l]; eB[s] = l == s ? 1.0 : 0.0; }
    // a lane that owns no active row hands round 0: cs and every coupling INTO such a row are 0 (inv_diag = 0)
    BRow bb = b;
    if (!bb.on) bb.cs = 0.0;
    // contact-normal rows in u = lambda / 2^33 (cn_rowNs): what a normal slot hands round is scaled up at the receivers, what a
    // normal lane receives is scaled down; a friction row's bound mu * lambda_normal becomes (mu 2^33) * u_normal
    const double Sn = kNormalScale, iSn = 1.0 / kNormalScale;
#pragma unroll
    for (int s = 0; s < kNGen; s++) { nAB[s] *= Sn; nBB[s] *= Sn; }
    if (l < kNGen) {
#pragma unroll
        for (int j = 0; j < kNArows; j++) nBA[j] *= iSn;
#pragma unroll
        for (int s = 0; s < kNB; s++) nBB[s] *= iSn;
        bb.cs *= iSn;
    }
    bb.mu *= Sn;
    // the button's rows ride on motor rows 0..2 (cn_rowA2): one restart mask per pair
    const double e0 = eA[0] + eA[kBM], e1 = eA[1] + eA[kBLo], e2 = eA[2] + eA[kBHi];
    double accB = 0.0, uA = 0.0, lam = 0.0;
    switch (ngen_w) {                                  // wave-uniform
        case 0: lam = cn_sweeps<0>(r, bb, nBA, eA, e0, e1, e2, nAB, nBB, eB, accA, accB, uA, l); break;
        case 1: lam = cn_sweeps<1>(r, bb, nBA, eA, e0, e1, e2, nAB, nBB, eB, accA, accB, uA, l); break;
        case 2: lam = cn_sweeps<2>(r, bb, nBA, eA, e0, e1, e2, nAB, nBB, eB, accA, accB, uA, l); break;
        case 3: lam = cn_sweeps<3>(r, bb, nBA, eA, e0, e1, e2, nAB, nBB, eB, accA, accB, uA, l); break;
        case 4: lam = cn_sweeps<4>(r, bb, nBA, eA, e0, e1, e2, nAB, nBB, eB, accA, accB, uA, l); break;
        case 5: lam = cn_sweeps<5>(r, bb, nBA, eA, e0, e1, e2, nAB, nBB, eB, accA, accB, uA, l); break;
        default: lam = cn_sweeps<kNGen>(r, bb, nBA, eA, e0, e1, e2, nAB, nBB, eB, accA, accB, uA, l); break;
    }
    b.lam = lam * (l < kNGen ? Sn : 1.0);
    return uA;
}

// ------------------------------------------------------------------ PGS under the model's solver details (srlhip_kuka_tree_model.solver_detail)
// Any row ORDER — alternating sweep direction, body-creation order — and the second friction direction in ONE formulation: every
// lane keeps the TOTAL coupling sum of its rows over the current values of all other rows, tot_r = sum_k n_rk u_k, and a row update
// hands round its CHANGE:  t = clamp(cs_r + tot_r);  d = t - u_r;  u_r = t;  tot_i += n_ir d on every lane i.  (The default path's
// accumulators restart at the own row, which is only right when every other row is visited exactly once between two visits of a
// row: an alternating order breaks that.)  Three dependent float64 operations per row instead of two, the same rows, bounds and
// start values (lambda = 0, i.e. u = 1/2 on the symmetric rows); with solver_detail = 0 none of this code runs.
struct DState { double totA, uA, totB, totC, uC; };
template <int J, bool GEN> SRL_G void d_rowA(const TRows &r, const double *sc, int l, DState &st) {
    const double t = clamp01(r.cs + st.totA);
    const double d = t - st.uA;
    if (l == J) st.uA = t;
    fmac_bcast<J>(st.totA, d, r.n[J]);
    if constexpr (GEN) fmac_bcast<J>(st.totB, d, sc[SC_NBA + J * GL + l]);
}
// Kuka2Button: row J (kBM / kBLo / kBHi) of the second button, same lanes, own sums
template <int J, bool GEN> SRL_G void d_rowC(const TRows2 &r2, const double nBC[3], int l, DState &st) {
    const double t = clamp01(r2.cs + st.totC);
    const double d = t - st.uC;
    if (l == J) st.uC = t;
    fmac_bcast<J>(st.totC, d, r2.n[J - kBM]);
    if constexpr (GEN) fmac_bcast<J>(st.totB, d, nBC[J - kBM]);
}
// bank-B slot s (a wave-uniform index): see gen_rowB for the friction bounds and `part`
template <int NB, int RB = 0> SRL_G void d_rowB(const double *sc, int l, int s, BRow &b, DState &st, bool part, double nCB_s, RbRows *rr = nullptr) {
    double lo = b.lo, hi = b.hi;
    const double tot = shfl(b.lam, b.normal);
    const bool skip = b.fric && !(tot > 0.0);
    if (b.fric) { lo = -b.mu * tot; hi = b.mu * tot; }
    double t = b.cs + st.totB;
    if constexpr (RB) {          // the body's current velocity change along the row (dv space: it lives on the body's lane)
        const int ob = b.obj >= 0 ? b.obj : 0;
        const double jo = b.Jo[0] * shfl(rr->dv[0], ob) + b.Jo[1] * shfl(rr->dv[1], ob) + b.Jo[2] * shfl(rr->dv[2], ob);
        // (the total-sum form excludes the row's own impulse from its residual: the body's dv carries it, so it is taken out again)
        if (b.obj >= 0) t -= (jo - b.jo2m * b.lam) * b.inv_diag;
    }
    t = t < lo ? lo : (t > hi ? hi : t);
    if (skip) t = b.lam;
    const double db = shfl(part ? t - b.lam : 0.0, s);
    st.totA = fma(sc[SC_NAB + s * GL + l], db, st.totA);
    st.totB = fma(sc[SC_NBB + s * GL + l], db, st.totB);     // (the plane holds 0 at [s][s])
    if (part && l == s) b.lam = t;
    if constexpr (NB == 2) st.totC = fma(nCB_s, db, st.totC);
    if constexpr (RB) {          // ... and the row's impulse change moves it
        const int ob = (int)shfl((double)b.obj, s);
        const double j0 = shfl(b.Jo[0], s), j1 = shfl(b.Jo[1], s), j2 = shfl(b.Jo[2], s);
        if (l == ob) { rr->dv[0] += db * j0 / kRbMass; rr->dv[1] += db * j1 / kRbMass; rr->dv[2] += db * j2 / kRbMass; }
    }
}
// The non-contact rows of one sweep in the order the details ask for.  Creation order of the solver (detail bit 1 clear): motors 0..11,
// button motor(s), joint limits, button stops (b = 0: lower, upper; b = 1: ...).  Body order (bit 1): per button its stops, then its
// motor; the arm's limits; the arm's motors.  REV: the same list backwards.  LIM(FWD) sweeps the limit slots.
#define SRL_D_ARM_FWD  d_rowA<0, GEN>(r, sc, l, st); d_rowA<1, GEN>(r, sc, l, st); d_rowA<2, GEN>(r, sc, l, st); d_rowA<3, GEN>(r, sc, l, st);   \
                       d_rowA<4, GEN>(r, sc, l, st); d_rowA<5, GEN>(r, sc, l, st); d_rowA<6, GEN>(r, sc, l, st); d_rowA<7, GEN>(r, sc, l, st);   \
                       d_rowA<8, GEN>(r, sc, l, st); d_rowA<9, GEN>(r, sc, l, st); d_rowA<10, GEN>(r, sc, l, st); d_rowA<11, GEN>(r, sc, l, st);
#define SRL_D_ARM_REV  d_rowA<11, GEN>(r, sc, l, st); d_rowA<10, GEN>(r, sc, l, st); d_rowA<9, GEN>(r, sc, l, st); d_rowA<8, GEN>(r, sc, l, st); \
                       d_rowA<7, GEN>(r, sc, l, st); d_rowA<6, GEN>(r, sc, l, st); d_rowA<5, GEN>(r, sc, l, st); d_rowA<4, GEN>(r, sc, l, st);   \
                       d_rowA<3, GEN>(r, sc, l, st); d_rowA<2, GEN>(r, sc, l, st); d_rowA<1, GEN>(r, sc, l, st); d_rowA<0, GEN>(r, sc, l, st);
template <int NB, bool GEN, class LIM>
SRL_G void d_noncontact(const TRows &r, const TRows2 &r2, const double nBC[3], const double *sc, int l, DState &st, bool body_order, bool rev, LIM lim) {
    if (!body_order) {
        if (!rev) {
            SRL_D_ARM_FWD
            d_rowA<kBM, GEN>(r, sc, l, st);
            if constexpr (NB == 2) d_rowC<kBM, GEN>(r2, nBC, l, st);
            if constexpr (GEN) lim(true);
            d_rowA<kBLo, GEN>(r, sc, l, st); d_rowA<kBHi, GEN>(r, sc, l, st);
            if constexpr (NB == 2) { d_rowC<kBLo, GEN>(r2, nBC, l, st); d_rowC<kBHi, GEN>(r2, nBC, l, st); }
        } else {
            if constexpr (NB == 2) { d_rowC<kBHi, GEN>(r2, nBC, l, st); d_rowC<kBLo, GEN>(r2, nBC, l, st); }
            d_rowA<kBHi, GEN>(r, sc, l, st); d_rowA<kBLo, GEN>(r, sc, l, st);
            if constexpr (GEN) lim(false);
            if constexpr (NB == 2) d_rowC<kBM, GEN>(r2, nBC, l, st);
            d_rowA<kBM, GEN>(r, sc, l, st);
            SRL_D_ARM_REV
        }
    } else {
        if (!rev) {
            d_rowA<kBLo, GEN>(r, sc, l, st); d_rowA<kBHi, GEN>(r, sc, l, st); d_rowA<kBM, GEN>(r, sc, l, st);
            if constexpr (NB == 2) { d_rowC<kBLo, GEN>(r2, nBC, l, st); d_rowC<kBHi, GEN>(r2, nBC, l, st); d_rowC<kBM, GEN>(r2, nBC, l, st); }
            if constexpr (GEN) lim(true);
            SRL_D_ARM_FWD
        } else {
            SRL_D_ARM_REV
            if constexpr (GEN) lim(false);
            if constexpr (NB == 2) { d_rowC<kBM, GEN>(r2, nBC, l, st); d_rowC<kBHi, GEN>(r2, nBC, l, st); d_rowC<kBLo, GEN>(r2, nBC, l, st); }
            d_rowA<kBM, GEN>(r, sc, l, st); d_rowA<kBHi, GEN>(r, sc, l, st); d_rowA<kBLo, GEN>(r, sc, l, st);
        }
    }
}
#undef SRL_D_ARM_FWD
#undef SRL_D_ARM_REV
// start values: every impulse 0, i.e. u_k = -lo_k / S_k (1/2 on the symmetric rows, 0 on the unilateral ones)
template <class SOF, class LOF> SRL_G double d_u0(int k, SOF S_of, LOF lo_of) { const double Sk = S_of(k); return Sk > 0.0 ? -lo_of(k) / Sk : 0.0; }
// steps without generic rows: bank A (and the second button's rows) only
template <int NB, class SOF, class LOF>
SRL_G double sweeps_free_detail(const TRows &r, const TRows2 &r2, int detail, SOF S_of, LOF lo_of, double *u2_out) {
    int l = lane_id();
#if SRL_G_DEVICE
    asm volatile("" : "+v"(l));
#endif
    DState st;
    st.totA = 0.0; st.totB = 0.0; st.totC = 0.0;
#pragma unroll
    for (int k = 0; k < kNArows; k++) st.totA = fma(r.n[k], d_u0(k, S_of, lo_of), st.totA);
    st.uA = l < kNArows ? d_u0(l, S_of, lo_of) : 0.0;
    st.uC = 0.0;
    if constexpr (NB == 2) {
#pragma unroll
        for (int k = kBM; k <= kBHi; k++) st.totC = fma(r2.n[k - kBM], d_u0(k, S_of, lo_of), st.totC);
        st.uC = (l >= kBM && l <= kBHi) ? d_u0(l, S_of, lo_of) : 0.0;
    }
    const double nBC[3] = {0.0, 0.0, 0.0};
    const bool body_order = (detail & kDetailBodyOrder) != 0, alt = (detail & kDetailAltSweep) != 0;
    for (int it = 0; it < kSolverIters; it++)
        d_noncontact<NB, false>(r, r2, nBC, nullptr, l, st, body_order, alt && !(it & 1), [](bool) {});
    if constexpr (NB == 2) *u2_out = st.uC;
    return st.uA;
}

// the own collision sphere against the button's cap and base (the detection block of tphysics_step as a function: the OCC variant's
// general path recomputes it in its turn; culling is conservative, so the contact decisions are the same with any set of active rows)
SRL_G void tdetect(const TL &L, const GState &g, const Env &e, double cc[3], double n_cap[3], double n_base[3], double &d_cap, double &d_base) {
    const bool sphere = L.slink() >= 0;
    {
        double Rs[9], ps[3];
        link_frame(g, sphere ? L.slink() : 0, Rs, ps);
        const double sph3[3] = {L.sph(0), L.sph(1), L.sph(2)};
        frame_point(Rs, ps, sph3, cc);
    }
    const double cap_z0 = e.bz + kGliderOriginZ + e.bq;
    const double reach = L.sph(3) + kContactThreshold + 1e-9, dx = cc[0] - e.bx, dy = cc[1] - e.by, rho2 = dx * dx + dy * dy;
    const double rmax = kBaseRadius + reach;
    const double top = fmax(cap_z0 + kCapHeight, e.bz + kBaseHeight), bottom = fmin(cap_z0, e.bz);
    const bool far = cc[2] - top >= reach || bottom - cc[2] >= reach || rho2 >= rmax * rmax;
    if (wany(sphere && !far)) {
        if (sphere) {
            d_cap = sphere_cylinder(cc, L.sph(3), e.bx, e.by, kCapRadius, cap_z0, cap_z0 + kCapHeight, n_cap);
            d_base = sphere_cylinder(cc, L.sph(3), e.bx, e.by, kBaseRadius, e.bz, e.bz + kBaseHeight, n_base);
        }
    }
}

// ------------------------------------------------------------------ the general path
// Steps that carry joint-limit / contact / friction rows are rare (a few percent of the wavefront-steps): row definitions,
// couplings through LDS, the two-bank sweeps.  (Tried as a real call with its own frame, -mllvm -amdgpu-function-calls: the call
// site saves ~150 live registers to scratch, the throughput did not change, and one GPU parity test failed — inlined again.)
struct GenIn {
    const double *tab; double *scratch;
    TRows r;                      // the scaled bank-A row of this lane
    double qd_new, bqd, bound_bm;
    TRows2 r2; double bqd2;       // Kuka2Button: the second button's rows (same lanes), its glider velocity
    const RBody *rb;              // KukaRandButton: the own free body (velocities already carry gravity / the kick)
    double *park; const GState *g; const Env *e;      // OCC: the env's park area; the state the candidates are recomputed from
};
// What the general path needs of the step's intermediate results is parked in LDS when it is computed (planes the general path
// only overwrites at the end of its setup), so that nothing of it stays in registers on the common path:
//   W (the own row of M^-1) in the NBA plane [k][lane], S in its own plane, and MISC[13][lane] in the NAB / NBB planes:
//   sphere centre cc[3], n_cap[3], n_base[3], d_cap, d_base, pen_lo, pen_hi
constexpr int SC_STASH_W = SC_NBA, SC_STASH_MISC = SC_NAB;
enum { SM_CC = 0, SM_NCAP = 3, SM_NBASE = 6, SM_DCAP = 9, SM_DBASE, SM_PENLO, SM_PENHI, SM_COUNT,
       SM_NCAP2 = SM_COUNT, SM_NBASE2 = SM_NCAP2 + 3, SM_DCAP2 = SM_NBASE2 + 3, SM_DBASE2, SM_COUNT2 };     // Kuka2Button: the second button's shapes
static_assert(NJ * GL <= kNArows * GL && SM_COUNT2 * GL <= 2 * kNB * GL, "stash planes");
struct GenOut { double u, acc_b, dvb_b; double u2, dvb_b2; bool bodies_done; double dvo[3]; };   // + KukaRandButton: the own body's velocity change when its rows were swept here   // own bank-A value; sum_s nAB_s lambda_s; sum_s jb_s lambda_s / m of the bank-B rows (per glider)
// DET >= 0: the table's solver_detail as a compile-time constant (the configuration-specialised rollout instantiation: the host checks
// the installed table), -1: read from the lane table
// PART (the persistent kernels' early setup, tphysics_pre2 / tphysics_post2): 0 = the whole path; 1 = the setup only — row
// definitions, couplings in LDS, the own bank-B row: everything in front of the sweeps, and all of it independent of the step's
// action —, its per-lane result handed out in `ctx`; 2 = the sweeps and the outputs on a setup a PART = 1 call left behind.
struct GenCtx { BRow b; int nlim_w, ngen_w; bool on_lim, on_con; };
template <int NB = 1, int RB = 0, int OCC = 0, int DET = -1, int PART = 0>
SRL_G GenOut general_path(const GenIn &in, GenCtx *ctx = nullptr) {
    static_assert(!OCC || (NB == 1 && RB == 0), "the two-wavefronts-per-SIMD variant covers the one-button envs");
    // Written for a SMALL register footprint, not for speed (the path is rare): every loop over joints / slots is rolled and works
    // on LDS-resident data, so that the common path's long-lived values are not pushed into scratch by this code's pressure.
    SRL_TSTAMP(8); SRL_TCOUNT(17);
    const double dt = kDt, inv_dt = 1.0 / kDt;
    const double *tab = in.tab;
    double *sc = in.scratch;
    const TL L = lane_view(tab);
    const TRows &r = in.r;
    const double wb = 1.0 / kCapMass, blim = kLimitMaxImpulse;
    const double qd_new = in.qd_new, bound_bm = in.bound_bm, bqd = in.bqd;
    const bool is_button = L.l == kBM || L.l == kBLo || L.l == kBHi;
    // bank-B slot layout: normals / limits 0..ng-1, friction rows ng + g [, second friction direction 2 ng + g] — 6 + 6, or 4 + 4 + 4
    const int detail = DET >= 0 ? DET : L.detail(), ng = L.ng(), nfd = (detail & kDetailFriction2) ? 2 : 1, nslots = (1 + nfd) * ng;
    const double lerp = L.limit_erp(), cerp = L.contact_erp(), slop = L.linear_slop();
    auto used_slot = [&](int s_, int ngw) -> bool { return s_ < nslots && (s_ >= 2 * ng ? s_ - 2 * ng : s_ >= ng ? s_ - ng : s_) < ngw; };
    const double *wpark = OCC ? in.park + PK_W : sc + SC_STASH_W, *spark = OCC ? in.park + PK_S : sc + SC_S;
    auto S_of = [&](int k) -> double { return k < NJ ? 2.0 * tab[LT_BOUND * GL + k] : k == kBM ? 2.0 * bound_bm : k < GL - 1 ? blim : 0.0; };
    auto lo_of = [&](int k) -> double { return k < NJ ? -tab[LT_BOUND * GL + k] : k == kBM ? -bound_bm : 0.0; };
    BRow b;
    b.cs = 0.0; b.lo = 0.0; b.hi = 0.0; b.mu = 0.0; b.lam = 0.0; b.jb = 0.0; b.inv_diag = 0.0; b.normal = L.l; b.on = false; b.fric = false;
    b.bsel = 0; b.nBC[0] = 0.0; b.nBC[1] = 0.0; b.nBC[2] = 0.0;
    b.obj = -1; b.Jo[0] = 0.0; b.Jo[1] = 0.0; b.Jo[2] = 0.0; b.jo2m = 0.0;
    int nlim = 0, ngen = 0, nlim_w = 0, ngen_w = 0;
    bool on_lim = false, on_con = false;           // the own bank-B row belongs to the limit phase / the contact phase of the sweep
    if constexpr (PART == 2) { b = ctx->b; nlim_w = ctx->nlim_w; ngen_w = ctx->ngen_w; on_lim = ctx->on_lim; on_con = ctx->on_con; }
    if constexpr (PART != 2) {
    {
        // ---- the parked inputs of this lane (the MISC plane is reused for the NAB couplings below)
        double cc[3], n_cap[3] = {0, 0, 1}, n_base[3] = {0, 0, 1}, d_cap = 1e30, d_base = 1e30, pen_lo, pen_hi;
        if constexpr (OCC) {               // recomputed from the (unchanged) start-of-step state: same operations as tphysics_step's detection
            const GState &g = *in.g; const Env &e = *in.e;
            tdetect(L, g, e, cc, n_cap, n_base, d_cap, d_base);
            pen_lo = g.q - L.jlo(); pen_hi = L.jhi() - g.q;
        } else {
#pragma unroll
            for (int k = 0; k < 3; k++) { cc[k] = sc[SC_STASH_MISC + (SM_CC + k) * GL + L.l]; n_cap[k] = sc[SC_STASH_MISC + (SM_NCAP + k) * GL + L.l]; n_base[k] = sc[SC_STASH_MISC + (SM_NBASE + k) * GL + L.l]; }
            d_cap = sc[SC_STASH_MISC + SM_DCAP * GL + L.l]; d_base = sc[SC_STASH_MISC + SM_DBASE * GL + L.l];
            pen_lo = sc[SC_STASH_MISC + SM_PENLO * GL + L.l]; pen_hi = sc[SC_STASH_MISC + SM_PENHI * GL + L.l];
        }
        const bool sphere = L.slink() >= 0, has_lim = L.jnt && L.jlo() <= L.jhi();
        const bool c_cap = sphere && d_cap < kContactThreshold, c_base = sphere && d_base < kContactThreshold;
        double n_cap2[3] = {0, 0, 1}, n_base2[3] = {0, 0, 1}, d_cap2 = 1e30, d_base2 = 1e30;
        if constexpr (NB == 2) {
#pragma unroll
            for (int k = 0; k < 3; k++) { n_cap2[k] = sc[SC_STASH_MISC + (SM_NCAP2 + k) * GL + L.l]; n_base2[k] = sc[SC_STASH_MISC + (SM_NBASE2 + k) * GL + L.l]; }
            d_cap2 = sc[SC_STASH_MISC + SM_DCAP2 * GL + L.l]; d_base2 = sc[SC_STASH_MISC + SM_DBASE2 * GL + L.l];
        }
        const bool c_cap2 = NB == 2 && sphere && d_cap2 < kContactThreshold, c_base2 = NB == 2 && sphere && d_base2 < kContactThreshold;
        // KukaRandButton: the free body this lane's sphere touches (parked in the second button's slots: the env has one button)
        double n_obj[3] = {0, 0, 1}, d_obj = 1e30; int k_obj = -1;
        if constexpr (RB) {
#pragma unroll
            for (int k = 0; k < 3; k++) n_obj[k] = sc[SC_STASH_MISC + (SM_NCAP2 + k) * GL + L.l];
            d_obj = sc[SC_STASH_MISC + SM_DCAP2 * GL + L.l]; k_obj = (int)sc[SC_STASH_MISC + SM_DBASE2 * GL + L.l];
        }
        const bool c_obj = RB && sphere && k_obj >= 0;
        const bool lim_lo = has_lim && pen_lo <= kLimitActivationVel * dt, lim_hi = has_lim && pen_hi <= kLimitActivationVel * dt;
        // slot of a candidate = number of candidates before it in creation order: limits (joint 0 lower, joint 0 upper, joint 1
        // lower, ...), then contacts (sphere 0 cap, sphere 0 base, [sphere 0 cap 2, sphere 0 base 2,] sphere 1 cap, ...); the first max_gen are kept
        const uint32_t b_lo = ballot(lim_lo), b_hi = ballot(lim_hi), b_cap = ballot(c_cap), b_base = ballot(c_base);
        const uint32_t b_cap2 = NB == 2 ? ballot(c_cap2) : RB ? ballot(c_obj) : 0u, b_base2 = NB == 2 ? ballot(c_base2) : 0u;   // (RB: the object candidates take the cap-2 position: cap, base, body per sphere)
        const uint32_t below = (1u << L.l) - 1u;
        const int max_gen = L.max_gen();
        nlim = __builtin_popcount(b_lo) + __builtin_popcount(b_hi);
        const int ncon = __builtin_popcount(b_cap) + __builtin_popcount(b_base) + __builtin_popcount(b_cap2) + __builtin_popcount(b_base2);
        const int s_lo = __builtin_popcount(b_lo & below) + __builtin_popcount(b_hi & below), s_hi = s_lo + (lim_lo ? 1 : 0);
        const int s_cap = nlim + __builtin_popcount(b_cap & below) + __builtin_popcount(b_base & below) + __builtin_popcount(b_cap2 & below) + __builtin_popcount(b_base2 & below);
        const int s_base = s_cap + (c_cap ? 1 : 0), s_cap2 = s_base + (c_base ? 1 : 0), s_base2 = s_cap2 + (c_cap2 ? 1 : 0);
        if (nlim > max_gen) nlim = max_gen;
        ngen = nlim + ncon; if (ngen > max_gen) ngen = max_gen;
        // ---- row definitions -> LDS.  Slot s < kNGen: J[12] + (Jb, desired, position error, upper bound, on); its friction row at
        // slot kNGen + s: J[12] + (Jb, -, -, -, on, mu).  Every lane first clears the definition of its own slot.
        if (L.l < kNB) {
            double *d = sc + SC_DEF + L.l * kDefDoubles;
#pragma unroll
            for (int k = 0; k < kDefDoubles; k++) d[k] = 0.0;
        }
        sync_scratch();                    // (also: every lane has read its parked MISC values)
        auto put_limit = [&](int slot, double sign, double pen) {
            if (slot < max_gen) {
                double *o = sc + SC_J + slot * NJ, *d = sc + SC_DEF + slot * kDefDoubles;
#pragma nounroll
                for (int j = 0; j < NJ; j++) o[j] = j == L.l ? sign : 0.0;
                d[0] = 0.0; d[1] = pen > 0 ? -pen * inv_dt : 0.0; d[2] = pen > 0 ? 0.0 : -pen * lerp * inv_dt; d[3] = blim; d[4] = 1.0;
            }
        };
        auto put_contact = [&](int slot, const double nrm[3], double dist, bool cap, int bsel, int obj = -1) {
            if (slot < max_gen) {
                double *o = sc + SC_J + slot * NJ, *of = sc + SC_J + (ng + slot) * NJ;
                double *d = sc + SC_DEF + slot * kDefDoubles, *df = sc + SC_DEF + (ng + slot) * kDefDoubles;
                double *of2 = sc + SC_J + (nfd == 2 ? 2 * ng + slot : ng + slot) * NJ, *df2 = sc + SC_DEF + (nfd == 2 ? 2 * ng + slot : ng + slot) * kDefDoubles;
                double pt3[3], tdir[3], tdir2[3];
                const double pen = dist + slop;                           // Bullet: penetration = distance + m_linearSlop
                const double rad = L.sph(3), smu = L.smu();
                const uint32_t sanc = L.sanc();
#pragma unroll
                for (int k = 0; k < 3; k++) pt3[k] = cc[k] - rad * nrm[k];
                // btPlaneSpace1: first tangent of the contact normal (the one friction direction of Bullet's multibody solver)
                if (fabs(nrm[2]) > 0.7071067811865475244) { const double a = nrm[1] * nrm[1] + nrm[2] * nrm[2], kk = 1.0 / sqrt(a); tdir[0] = 0.0; tdir[1] = -nrm[2] * kk; tdir[2] = nrm[1] * kk; }
                else { const double a = nrm[0] * nrm[0] + nrm[1] * nrm[1], kk = 1.0 / sqrt(a); tdir[0] = -nrm[1] * kk; tdir[1] = nrm[0] * kk; tdir[2] = 0.0; }
                cross3(nrm, tdir, tdir2);                                 // the second friction direction (SOLVER_USE_2_FRICTION_DIRECTIONS)
#pragma unroll 4
                for (int j = 0; j < NJ; j++) {
                    const double *Sj = spark + j * 6;
                    const double Swj[3] = {Sj[0], Sj[1], Sj[2]}, Svj[3] = {Sj[3], Sj[4], Sj[5]};
                    double c3[3];
                    cross3(Swj, pt3, c3);                                 // w_j x pt + v_j = velocity of the contact point per unit qd_j
                    const double on = (sanc >> j) & 1u ? 1.0 : 0.0;       // only the joints the sphere's link hangs on
                    o[j] = on * (dot3(nrm, c3) + dot3(nrm, Svj));
                    of[j] = on * (dot3(tdir, c3) + dot3(tdir, Svj));
                    if (nfd == 2) of2[j] = on * (dot3(tdir2, c3) + dot3(tdir2, Svj));
                }
                d[0] = cap ? -nrm[2] : 0.0; d[1] = pen > 0 ? -pen * inv_dt : 0.0; d[2] = pen > 0 ? 0.0 : -pen * cerp * inv_dt; d[3] = 1e10; d[4] = 1.0;
                df[0] = cap ? -tdir[2] : 0.0; df[4] = (L.friction() && smu > 0.0) ? 1.0 : 0.0; df[5] = smu;
                d[6] = (double)bsel; df[6] = (double)bsel;
                if (nfd == 2) { df2[0] = cap ? -tdir2[2] : 0.0; df2[4] = df[4]; df2[5] = smu; df2[6] = (double)bsel; }
                if constexpr (RB) {      // body index + 1 in every row of the contact; the contact normal in the friction definitions' free slots
                    d[7] = (double)(obj + 1); df[7] = (double)(obj + 1); df[1] = nrm[0]; df[2] = nrm[1]; df[3] = nrm[2];
                    if (nfd == 2) { df2[7] = (double)(obj + 1); df2[1] = nrm[0]; df2[2] = nrm[1]; df2[3] = nrm[2]; }
                }
            }
        };
        if (lim_lo) put_limit(s_lo, 1.0, pen_lo);
        if (lim_hi) put_limit(s_hi, -1.0, pen_hi);
        if (c_cap) put_contact(s_cap, n_cap, d_cap, true, 0);
        if (c_base) put_contact(s_base, n_base, d_base, false, 0);
        if constexpr (NB == 2) {
            if (c_cap2) put_contact(s_cap2, n_cap2, d_cap2, true, 1);
            if (c_base2) put_contact(s_base2, n_base2, d_base2, false, 1);
        }
        if constexpr (RB) { if (c_obj) put_contact(s_cap2, n_obj, d_obj, false, 0, k_obj); }
        sync_scratch();
    }
    SRL_TSTAMP(12);                         // candidates -> row definitions in LDS
    nlim_w = 0; ngen_w = 0;
#pragma nounroll
    for (int k = 0; k < kNGen; k++) { if (wany(k < nlim)) nlim_w = k + 1; if (wany(k < ngen)) ngen_w = k + 1; }
    // ---- W J of every active slot: joint lane k computes (W J_s)_k = its coupling a_{k,s} (W: the parked row of M^-1 in the NBA
    //      plane); button lanes: jb wb Jb_s.  The own bank-A row's scaled couplings -a / (a_rr S_r) go straight to the NAB plane.
    {
        const bool liveA = r.S > 0.0 && r.diag > 0.0;
        const double invA = liveA ? 1.0 / (r.diag * r.S) : 0.0;
#pragma unroll 2
        for (int s = 0; s < kNB; s++) {
            const bool used = used_slot(s, ngen_w);
            double wjk = 0.0;
            if (used) {
                const double *Js = sc + SC_J + s * NJ, *ds = sc + SC_DEF + s * kDefDoubles;
                const double act = ds[4], jbs = ds[0];
#pragma unroll
                for (int j = 0; j < NJ; j++) wjk = fma(wpark[j * GL + L.l], act != 0.0 ? Js[j] : 0.0, wjk);   // (a slot this env does not use holds stale LDS: select, never multiply by 0)
                if (L.jnt) sc[SC_WJ + s * NJ + L.l] = wjk;
                else wjk = (is_button && act != 0.0 && (NB == 1 || ds[6] == 0.0)) ? r.jb * wb * jbs : 0.0;
            }
            sc[SC_NAB + s * GL + L.l] = -wjk * invA;
        }
    }
    sync_scratch();                        // W J complete; the parked W rows (NBA plane) are dead from here on
    SRL_TSTAMP(13);                         // W J of every slot
    {
        // ---- the own bank-B row (slot == lane): scalars, diagonal, couplings to bank A and to bank B
        const int own_slot = L.l < kNB ? L.l : 0;                       // lanes >= kNB own no bank-B row
        const double *myd = sc + SC_DEF + own_slot * kDefDoubles;
        const bool own_on = L.l < kNB && myd[4] != 0.0;
        const bool mine_f = own_on && L.l >= ng;
        const double own_jb = own_on ? myd[0] : 0.0;
        b.on = own_on; b.fric = mine_f; b.jb = own_jb; b.mu = mine_f ? myd[5] : 0.0;
        b.normal = mine_f ? (L.l >= 2 * ng ? L.l - 2 * ng : L.l - ng) : L.l;
        b.lo = 0.0; b.hi = (own_on && !mine_f) ? myd[3] : 0.0;
        const int own_bsel = (NB == 2 && own_on && myd[6] != 0.0) ? 1 : 0;
        b.bsel = own_bsel;
        const double *Jr = sc + SC_J + own_slot * NJ, *wjr = sc + SC_WJ + own_slot * NJ;
        double diag = own_jb * own_jb * wb, jv = own_jb * (own_bsel ? in.bqd2 : bqd), offb = own_jb * wb * (-bound_bm);
        if constexpr (RB) {
            // the row's part on a free body: Jacobian -n (normal row), -t1 / -t2 (its friction rows); the normal sits in the first
            // friction definition of the contact
            const int obj = own_on ? (int)myd[7] - 1 : -1;
            const int gslot = L.l >= 2 * ng ? L.l - 2 * ng : L.l >= ng ? L.l - ng : L.l;
            const double *nd = sc + SC_DEF + ((gslot + ng) % kNB) * kDefDoubles;
            const double nrm[3] = {nd[1], nd[2], nd[3]};
            double dir[3] = {nrm[0], nrm[1], nrm[2]};
            if (L.l >= ng) {
                double t1[3];
                if (fabs(nrm[2]) > 0.7071067811865475244) { const double a = nrm[1] * nrm[1] + nrm[2] * nrm[2], kk = 1.0 / sqrt(a); t1[0] = 0.0; t1[1] = -nrm[2] * kk; t1[2] = nrm[1] * kk; }
                else { const double a = nrm[0] * nrm[0] + nrm[1] * nrm[1], kk = 1.0 / sqrt(a); t1[0] = -nrm[1] * kk; t1[1] = nrm[0] * kk; t1[2] = 0.0; }
                if (L.l >= 2 * ng) cross3(nrm, t1, dir); else { dir[0] = t1[0]; dir[1] = t1[1]; dir[2] = t1[2]; }
            }
            b.obj = obj;
            const int ob = obj >= 0 ? obj : 0;
            const double vo[3] = {shfl(in.rb->v[0], ob), shfl(in.rb->v[1], ob), shfl(in.rb->v[2], ob)};
            if (obj >= 0) {
#pragma unroll
                for (int k = 0; k < 3; k++) { b.Jo[k] = -dir[k]; b.jo2m += b.Jo[k] * b.Jo[k] / kRbMass; jv += b.Jo[k] * vo[k]; }
                diag += b.jo2m;
            }
        }
#pragma unroll 4
        for (int j = 0; j < NJ; j++) {
            const double Jj = own_on ? Jr[j] : 0.0, wj = own_on ? wjr[j] : 0.0;
            const double qj = shfl(qd_new, j);                           // the unconstrained velocity of joint j
            diag = fma(Jj, wj, diag); jv = fma(Jj, qj, jv); offb = fma(wj, lo_of(j), offb);
        }
        const bool live = own_on && diag > 0.0;
        b.inv_diag = live ? 1.0 / diag : 0.0;
        if (!live) b.on = false;
        // (desired - J v + position term) / a_rr, minus what the bank-A rows contribute at their lower bounds
        const double des = (own_on && !mine_f) ? myd[1] : 0.0, perr = (own_on && !mine_f) ? myd[2] : 0.0;
        b.cs = ((des - jv) + perr - offb) * b.inv_diag;
        // couplings of the own bank-B row to the bank-A rows j (in u units: a_rj S_j) and to the bank-B rows s
#pragma unroll 5
        for (int j = 0; j < kNArows; j++) {
            double a = 0.0;
            if (j < NJ) a = own_on ? wjr[j] : 0.0;
            else if (j == kBM || j == kBLo) a = own_jb * wb;
            else if (j == kBHi) a = -own_jb * wb;
            if (NB == 2 && j >= NJ && own_bsel) a = 0.0;        // the row acts on ONE glider: its couplings go to that button's rows
            sc[SC_NBA + j * GL + L.l] = -a * S_of(j) * b.inv_diag;
        }
        if constexpr (NB == 2) {
            const double ab = own_bsel ? own_jb * wb * b.inv_diag : 0.0;
            b.nBC[0] = -ab * S_of(kBM); b.nBC[1] = -ab * S_of(kBLo); b.nBC[2] = ab * S_of(kBHi);
        }
#pragma unroll 2
        for (int s = 0; s < kNB; s++) {
            const bool used = used_slot(s, ngen_w);
            double a = 0.0;
            if (used && s != L.l) {
                const double *ws = sc + SC_WJ + s * NJ, *ds = sc + SC_DEF + s * kDefDoubles;
                a = (ds[4] != 0.0 && (NB == 1 || (ds[6] != 0.0) == (own_bsel != 0))) ? own_jb * wb * ds[0] : 0.0;
#pragma unroll
                for (int j = 0; j < NJ; j++) a = fma(own_on ? Jr[j] : 0.0, ds[4] != 0.0 ? ws[j] : 0.0, a);
            }
            sc[SC_NBB + s * GL + L.l] = -a * b.inv_diag;
        }
        on_lim = b.on && L.l < nlim; on_con = b.on && !on_lim;
    }
    sync_scratch();
    SRL_TSTAMP(14);                         // the own bank-B row: diagonal, right-hand side, couplings
    }       // PART != 2
    if constexpr (PART == 1) {
        ctx->b = b; ctx->nlim_w = nlim_w; ctx->ngen_w = ngen_w; ctx->on_lim = on_lim; ctx->on_con = on_con;
        GenOut none = {};
        return none;
    }
    // ---- Bullet's row order: motors 0..11, button motor, [joint limits], button stops, [contact normals], [frictions].
    // Every impulse starts at 0, i.e. u_k = -lo_k / S_k = 1/2 on the symmetric bank-A rows (motors, button motor: all swept before
    // any bank-B row): row l starts with what the bank-A rows BEHIND it contribute at that value.
    double accA = 0.0, accB = 0.0, uA = 0.0;
#pragma unroll
    for (int k = 0; k < kNArows; k++) {
        const double Sk = S_of(k), u0 = Sk > 0.0 ? -lo_of(k) / Sk : 0.0;
        accA = fma(r.n[k] * (k > L.l ? 1.0 : 0.0), u0, accA);
    }
    double accC = 0.0, uC = 0.0;
    // Kuka2Button: the second button's rows' scaled couplings to the bank-B slots that act on glider 2 (the form the first button's have in the NAB plane)
    const bool liveC = NB == 2 && is_button && in.r2.S > 0.0;
    const double invC = liveC ? 1.0 / (wb * in.r2.S) : 0.0;
    auto nCB_of = [&](int s) -> double {
        const bool used = used_slot(s, ngen_w);
        const double *ds = sc + SC_DEF + s * kDefDoubles;
        return (used && liveC && ds[4] != 0.0 && ds[6] != 0.0) ? -(in.r2.jb * wb * ds[0]) * invC : 0.0;
    };
    // KukaRandButton: a row of this wavefront acts on a free body -> the bodies' own rows are swept with the arm's (total-sum form:
    // it takes the body terms in dv space); otherwise the caller solves them in closed form
    const bool obj_rows_w = RB && wany(b.on && b.obj >= 0);
    RbRows rr;
    if constexpr (RB) { if (obj_rows_w) rb_rows_setup(*in.rb, L.table_z(), cerp, slop, L.friction(), rr); }
    if (nlim_w > 0) SRL_TCOUNT(18);        // profiling build: steps of this wavefront with a joint-limit row
    if (ngen_w > nlim_w) SRL_TCOUNT(19);   //                  ... with a contact row
    if (detail != 0 || obj_rows_w) {
        // the model's solver details: total-sum formulation (d_rowA / d_rowB above), rows in the order the bits ask for
        DState st;
        st.totA = 0.0; st.totB = 0.0; st.totC = 0.0; st.uC = 0.0;
#pragma unroll
        for (int k = 0; k < kNArows; k++) {
            const double u0 = d_u0(k, S_of, lo_of);
            st.totA = fma(r.n[k], u0, st.totA);
            st.totB = fma(sc[SC_NBA + k * GL + L.l], u0, st.totB);
        }
        st.uA = L.l < kNArows ? d_u0(L.l, S_of, lo_of) : 0.0;
        if constexpr (NB == 2) {
#pragma unroll
            for (int k = kBM; k <= kBHi; k++) { const double u0 = d_u0(k, S_of, lo_of); st.totC = fma(in.r2.n[k - kBM], u0, st.totC); st.totB = fma(b.nBC[k - kBM], u0, st.totB); }
            st.uC = is_button ? d_u0(L.l, S_of, lo_of) : 0.0;
        }
        const bool body_order = (detail & kDetailBodyOrder) != 0, alt = (detail & kDetailAltSweep) != 0;
        const int l = L.l;
        for (int it = 0; it < kSolverIters; it++) {
            d_noncontact<NB, true>(r, in.r2, b.nBC, sc, l, st, body_order, alt && !(it & 1), [&](bool fwd) {
                for (int k = 0; k < nlim_w; k++) { const int s = fwd ? k : nlim_w - 1 - k; d_rowB<NB, RB>(sc, l, s, b, st, shfl(on_lim ? 1.0 : 0.0, s) != 0.0, NB == 2 ? nCB_of(s) : 0.0, &rr); }
            });
            for (int s = 0; s < ngen_w; s++) d_rowB<NB, RB>(sc, l, s, b, st, shfl(on_con ? 1.0 : 0.0, s) != 0.0, NB == 2 ? nCB_of(s) : 0.0, &rr);
            if constexpr (RB) { if (obj_rows_w) rb_sweep_normal(rr); }       // the bodies' table-contact rows: behind the arm's normals ...
            for (int g = 0; g < ngen_w; g++)
                for (int f = 1; f <= nfd; f++) { const int s = f * ng + g; d_rowB<NB, RB>(sc, l, s, b, st, shfl(on_con ? 1.0 : 0.0, s) != 0.0, NB == 2 ? nCB_of(s) : 0.0, &rr); }
            if constexpr (RB) { if (obj_rows_w) rb_sweep_friction(rr); }     // ... and their friction rows behind the arm's
        }
        uA = st.uA; uC = st.uC;
    } else if (nlim_w == 0) {
        if constexpr (NB == 2) {
            double nCB[kNB];
#pragma unroll
            for (int s = 0; s < kNB; s++) nCB[s] = nCB_of(s);
            uA = sweeps_contacts2(r, in.r2, b, sc, accA, ngen_w, nCB, &uC);
        } else uA = sweeps_contacts(r, b, sc, accA, ngen_w);
    } else
    for (int it = 0; it < kSolverIters; it++) {
        gen_rowA<0>(L, r, sc, accA, accB, uA);  gen_rowA<1>(L, r, sc, accA, accB, uA);  gen_rowA<2>(L, r, sc, accA, accB, uA);
        gen_rowA<3>(L, r, sc, accA, accB, uA);  gen_rowA<4>(L, r, sc, accA, accB, uA);  gen_rowA<5>(L, r, sc, accA, accB, uA);
        gen_rowA<6>(L, r, sc, accA, accB, uA);  gen_rowA<7>(L, r, sc, accA, accB, uA);  gen_rowA<8>(L, r, sc, accA, accB, uA);
        gen_rowA<9>(L, r, sc, accA, accB, uA);  gen_rowA<10>(L, r, sc, accA, accB, uA); gen_rowA<11>(L, r, sc, accA, accB, uA);
        gen_rowA<kBM>(L, r, sc, accA, accB, uA);
        if constexpr (NB == 2) gen_rowC<kBM>(L, in.r2, b, accC, accB, uC);
        // a slot index is a limit row in one env of the wavefront and a contact row in another: `part` keeps every row in its phase
        for (int s = 0; s < nlim_w; s++) gen_rowB<NB>(L, sc, s, b, accA, accB, shfl(on_lim ? 1.0 : 0.0, s) != 0.0, &accC, NB == 2 ? nCB_of(s) : 0.0);
        gen_rowA<kBLo>(L, r, sc, accA, accB, uA); gen_rowA<kBHi>(L, r, sc, accA, accB, uA);
        if constexpr (NB == 2) { gen_rowC<kBLo>(L, in.r2, b, accC, accB, uC); gen_rowC<kBHi>(L, in.r2, b, accC, accB, uC); }
        for (int s = 0; s < ngen_w; s++) gen_rowB<NB>(L, sc, s, b, accA, accB, shfl(on_con ? 1.0 : 0.0, s) != 0.0, &accC, NB == 2 ? nCB_of(s) : 0.0);
        for (int s = kNGen; s < kNGen + ngen_w; s++) gen_rowB<NB>(L, sc, s, b, accA, accB, shfl(on_con ? 1.0 : 0.0, s) != 0.0, &accC, NB == 2 ? nCB_of(s) : 0.0);
    }
    SRL_TSTAMP(15);                         // the 150 sweeps
    GenOut out;
    out.u = uA; out.acc_b = 0.0; out.dvb_b = 0.0; out.u2 = uC; out.dvb_b2 = 0.0;
    out.bodies_done = obj_rows_w; out.dvo[0] = 0.0; out.dvo[1] = 0.0; out.dvo[2] = 0.0;
    if constexpr (RB) { if (obj_rows_w) { out.dvo[0] = rr.dv[0]; out.dvo[1] = rr.dv[1]; out.dvo[2] = rr.dv[2]; } }
    const double pbb = b.on ? b.jb * b.lam * wb : 0.0;
    for (int s = 0; s < kNB; s++) {
        if (!used_slot(s, ngen_w)) continue;
        out.acc_b = fma(sc[SC_NAB + s * GL + L.l], shfl(b.on ? b.lam : 0.0, s), out.acc_b);
        out.dvb_b += shfl(b.bsel ? 0.0 : pbb, s);
        if constexpr (NB == 2) out.dvb_b2 += shfl(b.bsel ? pbb : 0.0, s);
    }
    sync_scratch();                          // scratch is reused by the next step
    SRL_TSTAMP(16);
    return out;
}

// ------------------------------------------------------------------ one physics step
// Kuka.applyAction (kuka.py:118-187) + p.stepSimulation() for the full model.  `e`: the env's scalar state replicated on the 16
// lanes, `g`: the lane's own joint and frame (valid on entry: trefresh()), jt_own: the joint-mode target of the own arm joint,
// finger_angle: motor_commands[4] (0.0 in every env of the reference: gripper closed).
// The action-independent half of a physics step: the own joint's spatial axis, then velocities / bias forces / composite inertias (the
// masked sums), the mass matrix (CRBA) and its inverse (Gauss-Jordan) -> the own row W of M^-1, the bias torque tau.  One definition for
// tphysics_step and for the persistent kernels, which run it for the NEXT step while they wait for the host's action (tphysics_pre).
SRL_G void tjoint_axis(const TL &L, const GState &g, double S[6]) {
    // ---- spatial joint axis about the world origin: S = [w ; p x w], w = R * axis
#pragma unroll
    for (int k = 0; k < 3; k++) S[k] = (g.R[k] * L.ax(0) + g.R[3 + k] * L.ax(1) + g.R[6 + k] * L.ax(2)) * L.jm;
    cross3(g.p, S, S + 3);
}
SRL_G void tdynamics(const TL &L, const GState &g, const double S[6], double qd, double W[NJ], double &tau) {
    {
        double w[3], vo[3], aw[3], av[3];
        {
            double le[NJ];                        // ancestors-or-self of the own link
            make_mask(L.anc(), le);
            {
                double xw[3], xv[3];
#pragma unroll
                for (int k = 0; k < 3; k++) { xw[k] = S[k] * qd; xv[k] = S[3 + k] * qd; }
                msum3(xw, le, w); msum3(xv, le, vo);
            }
            double t0[3], t1[3], t2[3];
            cross3(w, S, t0); cross3(w, S + 3, t1); cross3(vo, S, t2);
            {
                double xw[3], xv[3];
#pragma unroll
                for (int k = 0; k < 3; k++) { xw[k] = t0[k] * qd; xv[k] = (t1[k] + t2[k]) * qd; }
                msum3(xw, le, aw); msum3(xv, le, av, -kGravityZ);
            }
        }
        // rigid-body inertia of the own link about the world origin: Io (xx xy xz yy yz zz), h = m c
        double Io[6], h[3];
        {
            const double *R = g.R;
            double cw[3], T[9];
            const double com3[3] = {L.com(0), L.com(1), L.com(2)};
            frame_point(R, g.p, com3, cw);
            // T = R * Ilink (columns of T), Io = T * R^T + m (|c|^2 1 - c c^T)
            const double I00 = L.in(0), I01 = L.in(1), I02 = L.in(2), I11 = L.in(3), I12 = L.in(4), I22 = L.in(5);
#pragma unroll
            for (int k = 0; k < 3; k++) {
                T[k] = R[k] * I00 + R[3 + k] * I01 + R[6 + k] * I02;
                T[3 + k] = R[k] * I01 + R[3 + k] * I11 + R[6 + k] * I12;
                T[6 + k] = R[k] * I02 + R[3 + k] * I12 + R[6 + k] * I22;
            }
            const double m = L.mass(), ccs = dot3(cw, cw);
            Io[0] = T[0] * R[0] + T[3] * R[3] + T[6] * R[6] + m * (ccs - cw[0] * cw[0]);
            Io[1] = T[0] * R[1] + T[3] * R[4] + T[6] * R[7] - m * cw[0] * cw[1];
            Io[2] = T[0] * R[2] + T[3] * R[5] + T[6] * R[8] - m * cw[0] * cw[2];
            Io[3] = T[1] * R[1] + T[4] * R[4] + T[7] * R[7] + m * (ccs - cw[1] * cw[1]);
            Io[4] = T[1] * R[2] + T[4] * R[5] + T[7] * R[8] - m * cw[1] * cw[2];
            Io[5] = T[2] * R[2] + T[5] * R[5] + T[8] * R[8] + m * (ccs - cw[2] * cw[2]);
#pragma unroll
            for (int k = 0; k < 3; k++) h[k] = m * cw[k];
        }
        double Fn[3], Ff[3], Ioc[6], hc[3];
        {
            double n[3], f[3], t0[3], t1[3], fn[3], ff[3];
            sym_mul(Io, aw, n); cross3(h, av, t0); cross3(h, aw, t1);
#pragma unroll
            for (int k = 0; k < 3; k++) { fn[k] = n[k] + t0[k]; ff[k] = L.mass() * av[k] - t1[k]; }
            sym_mul(Io, w, n); cross3(h, vo, t0); cross3(h, w, t1);
#pragma unroll
            for (int k = 0; k < 3; k++) { n[k] += t0[k]; f[k] = L.mass() * vo[k] - t1[k]; }
            cross3(w, n, t0); cross3(vo, f, t1);
#pragma unroll
            for (int k = 0; k < 3; k++) fn[k] += t0[k] + t1[k];
            cross3(w, f, t0);
#pragma unroll
            for (int k = 0; k < 3; k++) ff[k] += t0[k];
            double ge[NJ];                        // descendants-or-self
            make_mask(L.desc(), ge);
            msum3(fn, ge, Fn); msum3(ff, ge, Ff); msum3(h, ge, hc);
            msum3(Io, ge, Ioc); msum3(Io + 3, ge, Ioc + 3);
        }
        tau = -L.damping() * qd - (dot3(S, Fn) + dot3(S + 3, Ff));
        SRL_TSTAMP(3);                      // velocities, bias forces, composite inertias (the masked sums)
        // ---- CRBA: M_kl = S_k . (Ic_l S_l) for k an ancestor-or-self of l (on lane l), mirrored; W = M^-1 in place
        double Fc[6], t0[3], t1[3], low[NJ];
        sym_mul(Ioc, S, Fc); cross3(hc, S + 3, t0); cross3(hc, S, t1);
#pragma unroll
        for (int k = 0; k < 3; k++) { Fc[k] += t0[k]; Fc[3 + k] = L.mcomp() * S[3 + k] - t1[k]; }
        dot6_all12(Fc, S, low);
        {
            double le[NJ];
            make_mask(L.anc(), le);
#pragma unroll
            for (int k = 0; k < NJ; k++) { low[k] *= le[k] * L.jm; W[k] = low[k]; }
        }
        transpose_step<0>(L, low, W);
        SRL_TSTAMP(4);                      // mass matrix (CRBA)
        double unused = 0.0;
        gj_step<0, NJ, true>(L, W, unused);
        SRL_TSTAMP(5);                      // its inverse (Gauss-Jordan)
    }
}
struct PreDyn { double S[6], W[NJ], tau; };
SRL_G void tphysics_pre(const GState &g, const double *tab, PreDyn &P) {
    const TL L = lane_view(tab);
    tjoint_axis(L, g, P.S);
    tdynamics(L, g, P.S, g.qd * L.jm, P.W, P.tau);
}

template <int NB = 1, int RB = 0, int OCC = 0, int DET = -1, int EARLY = 0>
SRL_G void tphysics_step(Env &e, GState &g, const double *tab, const Cfg &cfg, double *scratch, const double motor[3], bool joint_mode, double jt_own,
                         double finger_angle, RBody *rb = nullptr, double *park = nullptr, const PreDyn *pre = nullptr) {
    const double dt = kDt, inv_dt = 1.0 / kDt;
    const TL L = lane_view(tab);           // lane constants are read from LDS where they are used
    SRL_TSTAMP(0);                          // (everything between two physics steps: env logic, outputs, action sampling)
    double S[6];
    if constexpr (EARLY) {
#pragma unroll
        for (int k = 0; k < 6; k++) S[k] = pre->S[k];
    } else {
        tjoint_axis(L, g, S);
    }
    if (L.jnt) {                            // parked for the (rare) general path, see GenIn
#pragma unroll
        for (int k = 0; k < 6; k++) (OCC ? park + PK_S : scratch + SC_S)[L.l * 6 + k] = S[k];
    }
    // ---- IK target accumulate + clip (kuka.py:134-139), one damped-least-squares step on the arm block (kuka.py:144-156)
    double qdes = L.arm ? jt_own : L.tsel() * finger_angle;
    if (!joint_mode) {
        const int b = (cfg.random_target || cfg.two) ? 0 : 1;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            double v = e.ee[k] + motor[k];
            v = v < kEeBox[b][0][k] ? kEeBox[b][0][k] : v;
            v = v > kEeBox[b][1][k] ? kEeBox[b][1][k] : v;
            e.ee[k] = v;
        }
        double Rt[9], pt[3], ee[3], dS[6], J[6];
#pragma unroll
        for (int k = 0; k < 9; k++) Rt[k] = bcast<NA - 1>(g.R[k]);
#pragma unroll
        for (int k = 0; k < 3; k++) pt[k] = bcast<NA - 1>(g.p[k]);
        const double eept[3] = {L.eept(0), L.eept(1), L.eept(2)};
        frame_point(Rt, pt, eept, ee);
        {
            double d[3], Sa[3];
#pragma unroll
            for (int k = 0; k < 3; k++) { d[k] = ee[k] - g.p[k]; Sa[k] = S[k] * L.am; }
            cross3(Sa, d, J);
#pragma unroll
            for (int k = 0; k < 3; k++) J[3 + k] = Sa[k];
        }
#pragma unroll
        for (int k = 0; k < 3; k++) dS[k] = e.ee[k] - ee[k];
        {   // orientation error towards quat(euler(0, -pi, 0)), replicated (same construction as kuka_group.hpp)
            const double *R = Rt;
            double qx, qy, qz, qw;
            const double m00 = R[0], m01 = R[3], m02 = R[6], m10 = R[1], m11 = R[4], m12 = R[7], m20 = R[2], m21 = R[5], m22 = R[8];
            const double tr = m00 + m11 + m22;
            if (tr > 0) { double s = sqrt(tr + 1.0) * 2; qw = 0.25 * s; qx = (m21 - m12) / s; qy = (m02 - m20) / s; qz = (m10 - m01) / s; }
            else if (m00 > m11 && m00 > m22) { double s = sqrt(1.0 + m00 - m11 - m22) * 2; qw = (m21 - m12) / s; qx = 0.25 * s; qy = (m01 + m10) / s; qz = (m02 + m20) / s; }
            else if (m11 > m22) { double s = sqrt(1.0 + m11 - m00 - m22) * 2; qw = (m02 - m20) / s; qx = (m01 + m10) / s; qy = 0.25 * s; qz = (m12 + m21) / s; }
            else { double s = sqrt(1.0 + m22 - m00 - m11) * 2; qw = (m10 - m01) / s; qx = (m02 + m20) / s; qy = (m12 + m21) / s; qz = 0.25 * s; }
            const double tx = 0.0, ty = -1.0, tz = 0.0, tw = 6.123233995736766e-17;
            const double ix = -qx, iy = -qy, iz = -qz, iw = qw;
            const double dw = tw * iw - tx * ix - ty * iy - tz * iz;
            const double dx = tw * ix + tx * iw + ty * iz - tz * iy;
            const double dy = tw * iy - tx * iz + ty * iw + tz * ix;
            const double dz = tw * iz + tx * iy - ty * ix + tz * iw;
            const double sv = sqrt(dx * dx + dy * dy + dz * dz);
            double angle = 2.0 * atan2(sv, dw), ax, ay, az;
            if (sv * sv < 10.0 * 2.2204460492503131e-16) { ax = 1; ay = 0; az = 0; }
            else { ax = dx / sv; ay = dy / sv; az = dz / sv; }
            if (angle > kPi) angle -= 2 * kPi;
            dS[3] = angle * ax; dS[4] = angle * ay; dS[5] = angle * az;
        }
        double A[NA], bb = 0.0;
        dot6_step<0, NA>(J, J, A);
        const double damping = cfg.two ? kIkDampingDefault : kIkDamping;
#pragma unroll
        for (int k = 0; k < NA; k++) A[k] = fma(damping, L.e(k), A[k]);
#pragma unroll
        for (int c = 0; c < 6; c++) bb = fma(J[c], dS[c], bb);
        double det = 1.0;
        gj_step<0, NA, false>(L, A, bb, &det);
        if (det < kIkCrossDet) e.ikx |= 1;      // sticky until the episode's reset (kuka_core.hpp kIkCrossDet)
        bb *= L.am;
        double all[NA], maxabs = 0.0;
        ball_step<0, NA>(bb, all);
#pragma unroll
        for (int k = 0; k < NA; k++) maxabs = fmax(maxabs, fabs(all[k]));
        if (L.arm) qdes = g.q + bb;
        if (wany(maxabs > kIkMaxAngle)) {
            const double scale = kIkMaxAngle / maxabs;
            if (maxabs > kIkMaxAngle && L.arm) qdes = g.q + bb * scale;
        }
    }
    SRL_TSTAMP(1);                          // IK
    // ---- collision detection at the current poses: every lane owns one sphere of the model
    double cc[3], n_cap[3] = {0, 0, 1}, n_base[3] = {0, 0, 1}, d_cap = 1e30, d_base = 1e30;
    const bool sphere = L.slink() >= 0;
    {
        double Rs[9], ps[3];
        link_frame(g, sphere ? L.slink() : 0, Rs, ps);
        const double sph3[3] = {L.sph(0), L.sph(1), L.sph(2)};
        frame_point(Rs, ps, sph3, cc);
    }
    const double cap_z0 = e.bz + kGliderOriginZ + e.bq;
    {
        const double reach = L.sph(3) + kContactThreshold + 1e-9, dx = cc[0] - e.bx, dy = cc[1] - e.by, rho2 = dx * dx + dy * dy;
        const double rmax = kBaseRadius + reach;
        const double top = fmax(cap_z0 + kCapHeight, e.bz + kBaseHeight), bottom = fmin(cap_z0, e.bz);
        const bool far = cc[2] - top >= reach || bottom - cc[2] >= reach || rho2 >= rmax * rmax;
        if (wany(sphere && !far)) {
            if (sphere) {
                d_cap = sphere_cylinder(cc, L.sph(3), e.bx, e.by, kCapRadius, cap_z0, cap_z0 + kCapHeight, n_cap);
                d_base = sphere_cylinder(cc, L.sph(3), e.bx, e.by, kBaseRadius, e.bz, e.bz + kBaseHeight, n_base);
            }
        }
    }
    const bool c_cap = sphere && d_cap < kContactThreshold, c_base = sphere && d_base < kContactThreshold;
    if constexpr (!OCC) {
        double *m = scratch + SC_STASH_MISC + L.l;
#pragma unroll
        for (int k = 0; k < 3; k++) { m[(SM_CC + k) * GL] = cc[k]; m[(SM_NCAP + k) * GL] = n_cap[k]; m[(SM_NBASE + k) * GL] = n_base[k]; }
        m[SM_DCAP * GL] = d_cap; m[SM_DBASE * GL] = d_base;
    }
    bool c_any2 = false;
    if constexpr (NB == 2) {                // Kuka2Button: the second button's cap and base (same urdf, kuka_2button_gym_env.py:62-70)
        double n_cap2[3] = {0, 0, 1}, n_base2[3] = {0, 0, 1}, d_cap2 = 1e30, d_base2 = 1e30;
        const double cap2_z0 = e.bz + kGliderOriginZ + e.b2q;
        const double reach = L.sph(3) + kContactThreshold + 1e-9, dx = cc[0] - e.b2x, dy = cc[1] - e.b2y, rho2 = dx * dx + dy * dy;
        const double rmax = kBaseRadius + reach;
        const double top = fmax(cap2_z0 + kCapHeight, e.bz + kBaseHeight), bottom = fmin(cap2_z0, e.bz);
        const bool far = cc[2] - top >= reach || bottom - cc[2] >= reach || rho2 >= rmax * rmax;
        if (wany(sphere && !far)) {
            if (sphere) {
                d_cap2 = sphere_cylinder(cc, L.sph(3), e.b2x, e.b2y, kCapRadius, cap2_z0, cap2_z0 + kCapHeight, n_cap2);
                d_base2 = sphere_cylinder(cc, L.sph(3), e.b2x, e.b2y, kBaseRadius, e.bz, e.bz + kBaseHeight, n_base2);
            }
        }
        double *m = scratch + SC_STASH_MISC + L.l;
#pragma unroll
        for (int k = 0; k < 3; k++) { m[(SM_NCAP2 + k) * GL] = n_cap2[k]; m[(SM_NBASE2 + k) * GL] = n_base2[k]; }
        m[SM_DCAP2 * GL] = d_cap2; m[SM_DBASE2 * GL] = d_base2;
        c_any2 = sphere && (d_cap2 < kContactThreshold || d_base2 < kContactThreshold);
        e.contact_body1 = gany(c_cap || c_base) ? 1 : 0;            // getContactPoints(button_uid, kuka_uid): any link of that button
        e.contact_body2 = gany(c_any2) ? 1 : 0;
    }
    e.contact_table = gany(sphere && (cc[2] - L.sph(3) - L.table_z() < kContactThreshold)) ? 1 : 0;
    e.contact_button = gany(c_cap) ? 1 : 0;
    bool c_obj = false;
    if constexpr (RB) {
        // KukaRandButton: gravity on the own free body; the own sphere against the bodies (the lowest-numbered one in reach), only
        // while some sphere of the wavefront is low enough to reach the tallest body
        RBody &B = *rb;
        if (B.on) B.v[2] += dt * kGravityZ;
        double n_obj[3] = {0, 0, 1}, d_obj = 1e30;
        int k_obj = -1;
        const double zmax = L.table_z() + kRbMaxHeight + kContactThreshold + 1e-9;
        if (wany(sphere && cc[2] - L.sph(3) < zmax)) {
#pragma nounroll
            for (int k = 0; k < kRbN; k++) {
                const double code = shfl(B.on ? (double)B.type : -1.0, k);
                const double xk[3] = {shfl(B.x[0], k), shfl(B.x[1], k), shfl(B.x[2], k)};
                if (sphere && k_obj < 0 && code >= 0.0) {
                    double nn[3];
                    const double d = sphere_body(cc, L.sph(3), xk, (int)code, nn);
                    if (d < kContactThreshold) { k_obj = k; d_obj = d; n_obj[0] = nn[0]; n_obj[1] = nn[1]; n_obj[2] = nn[2]; }
                }
            }
        }
        c_obj = k_obj >= 0;
        double *m = scratch + SC_STASH_MISC + L.l;
#pragma unroll
        for (int k = 0; k < 3; k++) m[(SM_NCAP2 + k) * GL] = n_obj[k];
        m[SM_DCAP2 * GL] = d_obj; m[SM_DBASE2 * GL] = (double)k_obj;
    }
    // ---- motor target velocity of the own joint (btMultiBodyJointMotor, velocityGain 1, targetVelocity 0)
    double target = L.kp() * (qdes - g.q) * inv_dt;
    target = target > L.maxvel() ? L.maxvel() : target;
    target = target < -L.maxvel() ? -L.maxvel() : target;
    SRL_TSTAMP(2);                          // collision detection, motor targets
    // ---- dynamics in world coordinates
    const double qd = g.qd * L.jm;
    double W[NJ], tau;
    if constexpr (EARLY) {
#pragma unroll
        for (int k = 0; k < NJ; k++) W[k] = pre->W[k];
        tau = pre->tau;
    } else {
        tdynamics(L, g, S, qd, W, tau);
    }
#pragma unroll
    for (int k = 0; k < NJ; k++) (OCC ? park + PK_W : scratch + SC_STASH_W)[k * GL + L.l] = W[k];
    double qdd = 0.0;
    rdot_step<0, NJ>(qdd, W, tau);
#pragma unroll
    for (int k = 0; k < NJ; k++) SRL_GDBG(0, L.l * NJ + k, W[k]);
    SRL_GDBG(1, L.l, qdd); SRL_GDBG(2, L.l, tau); SRL_GDBG(3, L.l, qdes); SRL_GDBG(4, L.l, target);
    // (explicit: left to the contraction pass, WHICH product joins the add depends on how many uses `qd` has — the persistent kernels take
    //  W and tau from tphysics_pre and use `qd` only here, and got g.qd * jm + round(dt * qdd) where every other kernel has this)
    const double qd_new = fma(dt, qdd, qd);
    e.bqd += dt * kGravityZ;
    if constexpr (NB == 2) e.b2qd += dt * kGravityZ;
    // ---- bank-A rows (impulse space, A = J W J^T): motor row per joint lane, the button's three scalar rows
    const double wb = 1.0 / kCapMass, blim = kLimitMaxImpulse;
    const double bound_bm = e.motor_on ? kButtonMaxForce * dt : kDefaultMotorImpulse;
    const bool is_bm = L.l == kBM, is_blo = L.l == kBLo, is_bhi = L.l == kBHi, is_button = is_bm || is_blo || is_bhi;
    const double lerp = L.limit_erp();
    TRows r;
    double rhs = 0.0, off = 0.0;
#pragma unroll
    for (int k = 0; k < GL; k++) r.n[k] = 0.0;
    r.lo = 0.0; r.S = 0.0; r.jb = 0.0; r.diag = 0.0;
    // bounds of every bank-A row are known on every lane without communication: motor row k has lo = -bound_k, S = 2 bound_k (the
    // lane table), the button motor -+bound_bm, the button stops [0, blim]
    auto S_of = [&](int k) -> double { return k < NJ ? 2.0 * tab[LT_BOUND * GL + k] : k == kBM ? 2.0 * bound_bm : k < GL - 1 ? blim : 0.0; };
    auto lo_of = [&](int k) -> double { return k < NJ ? -tab[LT_BOUND * GL + k] : k == kBM ? -bound_bm : 0.0; };
    // unscaled coupling of the own bank-A row to bank-A row k
    auto a_of = [&](int k) -> double {
        if (k < NJ) return L.jnt ? W[k] * (1.0 - L.e(k)) : 0.0;
        if (k == kBM) return is_blo ? wb : is_bhi ? -wb : 0.0;
        if (k == kBLo) return is_bm ? wb : is_bhi ? -wb : 0.0;
        if (k == kBHi) return (is_bm || is_blo) ? -wb : 0.0;
        return 0.0;
    };
    if (L.jnt) {
#pragma unroll
        for (int k = 0; k < NJ; k++) r.diag = fma(L.e(k), W[k], r.diag);
        rhs = target - qd_new; r.lo = -L.bound(); r.S = 2.0 * L.bound();
    } else if (is_bm) {
        rhs = (e.motor_on ? kButtonKp * (kButtonTarget - e.bq) * inv_dt : 0.0) - e.bqd;
        r.lo = -bound_bm; r.S = 2.0 * bound_bm; r.jb = 1.0; r.diag = wb;
    } else if (is_blo) {
        const double pen = e.bq - kGliderLower;
        rhs = ((pen > 0 ? -pen * inv_dt : 0.0) - e.bqd) + (pen > 0 ? 0.0 : -pen * lerp * inv_dt);
        r.S = blim; r.jb = 1.0; r.diag = wb;
    } else if (is_bhi) {
        const double pen = kGliderUpper - e.bq;
        rhs = ((pen > 0 ? -pen * inv_dt : 0.0) + e.bqd) + (pen > 0 ? 0.0 : -pen * lerp * inv_dt);
        r.S = blim; r.jb = -1.0; r.diag = wb;
    }
#pragma unroll
    for (int k = 0; k < kNArows; k++) off = fma(a_of(k), lo_of(k), off);
    // ---- Kuka2Button: the second button's motor and stops, on the same three lanes
    TRows2 r2;
    r2.cs = 0.0; r2.n[0] = 0.0; r2.n[1] = 0.0; r2.n[2] = 0.0; r2.lo = 0.0; r2.S = 0.0; r2.jb = 0.0;
    if constexpr (NB == 2) {
        double rhs2 = 0.0, off2 = 0.0;
        if (is_bm) rhs2 = (e.motor_on ? kButtonKp * (kButtonTarget - e.b2q) * inv_dt : 0.0) - e.b2qd;
        else if (is_blo) { const double pen = e.b2q - kGliderLower; rhs2 = ((pen > 0 ? -pen * inv_dt : 0.0) - e.b2qd) + (pen > 0 ? 0.0 : -pen * lerp * inv_dt); }
        else if (is_bhi) { const double pen = kGliderUpper - e.b2q; rhs2 = ((pen > 0 ? -pen * inv_dt : 0.0) + e.b2qd) + (pen > 0 ? 0.0 : -pen * lerp * inv_dt); }
        if (is_button) { r2.lo = r.lo; r2.S = r.S; r2.jb = r.jb; }
#pragma unroll
        for (int k = kBM; k <= kBHi; k++) off2 = fma(a_of(k), lo_of(k), off2);
        const bool live2 = is_button && r2.S > 0.0;
        const double inv2 = live2 ? rcp(wb * r2.S) : 0.0;
        r2.cs = live2 ? (rhs2 - off2) * inv2 + (is_bm ? 0.5 : 0.0) : 0.0;
#pragma unroll
        for (int k = kBM; k <= kBHi; k++) r2.n[k - kBM] = -(a_of(k) * S_of(k)) * inv2;
    }
    // ---- generic rows: joint-limit candidates of the own joint, contact candidates of the own sphere
    const bool has_lim = L.jnt && L.jlo() <= L.jhi();
    const double pen_lo = g.q - L.jlo(), pen_hi = L.jhi() - g.q;
    const bool lim_lo = has_lim && pen_lo <= kLimitActivationVel * dt, lim_hi = has_lim && pen_hi <= kLimitActivationVel * dt;
    if constexpr (!OCC) { scratch[SC_STASH_MISC + SM_PENLO * GL + L.l] = pen_lo; scratch[SC_STASH_MISC + SM_PENHI * GL + L.l] = pen_hi; }
    const bool any_generic = wany(lim_lo || lim_hi || c_cap || c_base || c_any2 || c_obj);
    // ---- scale the bank-A rows to u in [0, 1]:  x_r = cs_r + sum_k n_rk u_k
    {
        const bool live = r.S > 0.0 && r.diag > 0.0;
        const double inv = live ? rcp(r.diag * r.S) : 0.0;
        r.cs = live ? (rhs - off) * inv + ((L.jnt || is_bm) ? 0.5 : 0.0) : 0.0;
#pragma unroll
        for (int k = 0; k < kNArows; k++) r.n[k] = -(a_of(k) * S_of(k)) * inv;
        // lambda starts at 0, i.e. u_k = -lo_k / S_k = 1/2 for the symmetric rows: what motor row l sees of the motor rows behind it
        // during the first sweep (the button motor couples to no row that comes before it)
        r.acc0 = 0.0;
        if (L.jnt) {
#pragma unroll
            for (int k = 0; k < NJ; k++) r.acc0 = fma(r.n[k] * (k > L.l ? 1.0 : 0.0), 0.5, r.acc0);
        }
    }
    double u, acc_b = 0.0, dvb_b = 0.0, u2 = 0.0, dvb_b2 = 0.0, dvo[3] = {0.0, 0.0, 0.0};
    bool bodies_done = false;
    const int detail = DET >= 0 ? DET : L.detail();
    SRL_TSTAMP(6);                          // row setup
    if (!any_generic) u = detail != 0 ? sweeps_free_detail<NB>(r, r2, detail, S_of, lo_of, &u2) : sweeps_free<NB>(r, &r2, &u2);
    else if constexpr (OCC) {
        // one work area per wavefront: the envs that carry generic rows take it in turns (one 16-lane row active at a time: every
        // cross-lane operation of the general path is row-local, its wave votes then see that row only); the others sweep their
        // bank-A rows meanwhile as on a free step
        const bool need = gany(lim_lo || lim_hi || c_cap || c_base);
        u = 0.0;
        if (!need) u = detail != 0 ? sweeps_free_detail<NB>(r, r2, detail, S_of, lo_of, &u2) : sweeps_free<NB>(r, &r2, &u2);
        const int myrow = grp::row_id();
#pragma nounroll
        for (int turn = 0; turn < grp::kRowsPerWave; turn++) {
            if (need && myrow == turn) {
                GenIn in;
                in.tab = tab; in.scratch = scratch; in.r = r; in.qd_new = qd_new; in.bqd = e.bqd; in.bound_bm = bound_bm;
                in.r2 = r2; in.bqd2 = 0.0; in.rb = nullptr; in.park = park; in.g = &g; in.e = &e;
                const GenOut out = general_path<NB, RB, OCC, DET>(in);
                u = out.u; acc_b = out.acc_b; dvb_b = out.dvb_b;
            }
        }
    } else {
        GenIn in;
        in.tab = tab; in.scratch = scratch; in.r = r; in.qd_new = qd_new; in.bqd = e.bqd; in.bound_bm = bound_bm;
        in.r2 = r2; in.bqd2 = NB == 2 ? e.b2qd : 0.0; in.rb = rb; in.park = nullptr; in.g = &g; in.e = &e;
        const GenOut out = general_path<NB, RB, 0, DET>(in);
        u = out.u; acc_b = out.acc_b; dvb_b = out.dvb_b; u2 = out.u2; dvb_b2 = out.dvb_b2;
        if constexpr (RB) { bodies_done = out.bodies_done; dvo[0] = out.dvo[0]; dvo[1] = out.dvo[1]; dvo[2] = out.dvo[2]; }
    }
    if constexpr (RB) {
        // the own free body: rows swept with the arm's above, or (no arm contact in the wavefront) their closed form — table normal,
        // then the two friction rows: orthogonal rows on one body, the first sweep is the fixed point
        RBody &B = *rb;
        if (!bodies_done) {
            RbRows rr;
            rb_rows_setup(B, L.table_z(), L.contact_erp(), L.linear_slop(), L.friction(), rr);
            rb_sweep_normal(rr); rb_sweep_friction(rr);
            dvo[0] = rr.dv[0]; dvo[1] = rr.dv[1]; dvo[2] = rr.dv[2];
        }
        if (B.on) {
#pragma unroll
            for (int k = 0; k < 3; k++) { B.v[k] += dvo[k]; B.x[k] += dt * B.v[k]; }
        }
    }
    if (any_generic) { SRL_TSTAMP(8); } else { SRL_TSTAMP(7); }      // the 150 sweeps: free path / steps with generic rows (setup included)
    const double lam = r.lo + r.S * u;
    SRL_GDBG(5, lane_id(), lam);
    // ---- velocity change: joint lane i gets sum_r a_ir lambda_r, the glider sum_r jb_r lambda_r / m
    double dv = 0.0, dvb = 0.0;
    {
        const double v = r.S > 0.0 ? lam / r.S : 0.0, pb = r.jb * lam * wb;
        double acc = 0.0;
#define SRL_ACC(K) fmac_bcast<K>(acc, v, r.n[K]);
        SRL_ACC(0) SRL_ACC(1) SRL_ACC(2) SRL_ACC(3) SRL_ACC(4) SRL_ACC(5) SRL_ACC(6) SRL_ACC(7) SRL_ACC(8) SRL_ACC(9) SRL_ACC(10) SRL_ACC(11)
#undef SRL_ACC
        dvb = bcast<kBM>(pb) + bcast<kBLo>(pb) + bcast<kBHi>(pb);
        acc += acc_b; dvb += dvb_b;
        dv = r.diag * (lam - r.S * acc);
    }
    // ---- semi-implicit Euler, refresh sin/cos, frames and the gripper position (a fresh copy of the lane constants: the one loaded
    //      at the top of the step is not kept live across the solver loop)
    if (lane_id() < NJ) { g.qd = qd_new + dv; g.q += dt * g.qd; }
    e.bqd += dvb;
    e.bq += dt * e.bqd;
    if constexpr (NB == 2) {
        const double pb2 = r2.jb * (r2.lo + r2.S * u2) * wb;
        e.b2qd += bcast<kBM>(pb2) + bcast<kBLo>(pb2) + bcast<kBHi>(pb2) + dvb_b2;
        e.b2q += dt * e.b2qd;
    }
    const TL L3 = lane_view(tab);
    trefresh(L3, g, e);
    SRL_TSTAMP(9);                          // velocity update, integration, sin / cos, forward kinematics
}

// ---- the persistent kernel of the reference's default configuration (SPEC instantiation: one button, no free bodies, one wavefront per
// SIMD) splits the physics step at the ACTION: tphysics_pre2 runs everything that depends on the state only — joint axes, dynamics,
// collision detection, the bank-A rows up to their right-hand sides, and on a step with generic rows the general path's whole SETUP
// (general_path<..., PART = 1>: 16 k cycles of the contact step that sets a 4096-env step's latency) — while the wavefront waits for
// the host; tphysics_post2 takes the action: IK, motor targets, right-hand sides, the sweeps, integration.  The same operations on
// the same operands as tphysics_step (the blocks are its own, in another order); `e` is not modified before the action is there (a
// park in between stores the state as it was).
struct PreStep {
    double S[6], W[NJ], tau, qd, qd_new, bqd_g, bound_bm, rhs_b, off, inv;
    TRows r;                      // everything but cs
    int contact_table, contact_button;
    bool live, any_generic;
    GenCtx ctx;
};
template <int DET>
SRL_G void tphysics_pre2(const Env &e, const GState &g, const double *tab, double *scratch, PreStep &P) {
    const double dt = kDt, inv_dt = 1.0 / kDt;
    const TL L = lane_view(tab);
    // step_command switches the button's motor on in front of the physics step — unless the action is `None` (kuka_env.hpp): the one
    // thing the rows take from the action besides the IK target.  Assumed on here; tenv_step falls back to tphysics_step for a
    // wavefront in which it is not (an env's `None` action right after its reset).
    constexpr int motor_on = 1;
    double *S = P.S, *W = P.W;
    tjoint_axis(L, g, S);
    if (L.jnt) {                            // parked for the (rare) general path, see GenIn
#pragma unroll
        for (int k = 0; k < 6; k++) (scratch + SC_S)[L.l * 6 + k] = S[k];
    }
    // ---- collision detection at the current poses: every lane owns one sphere of the model
    double cc[3], n_cap[3] = {0, 0, 1}, n_base[3] = {0, 0, 1}, d_cap = 1e30, d_base = 1e30;
    const bool sphere = L.slink() >= 0;
    {
        double Rs[9], ps[3];
        link_frame(g, sphere ? L.slink() : 0, Rs, ps);
        const double sph3[3] = {L.sph(0), L.sph(1), L.sph(2)};
        frame_point(Rs, ps, sph3, cc);
    }
    const double cap_z0 = e.bz + kGliderOriginZ + e.bq;
    {
        const double reach = L.sph(3) + kContactThreshold + 1e-9, dx = cc[0] - e.bx, dy = cc[1] - e.by, rho2 = dx * dx + dy * dy;
        const double rmax = kBaseRadius + reach;
        const double top = fmax(cap_z0 + kCapHeight, e.bz + kBaseHeight), bottom = fmin(cap_z0, e.bz);
        const bool far = cc[2] - top >= reach || bottom - cc[2] >= reach || rho2 >= rmax * rmax;
        if (wany(sphere && !far)) {
            if (sphere) {
                d_cap = sphere_cylinder(cc, L.sph(3), e.bx, e.by, kCapRadius, cap_z0, cap_z0 + kCapHeight, n_cap);
                d_base = sphere_cylinder(cc, L.sph(3), e.bx, e.by, kBaseRadius, e.bz, e.bz + kBaseHeight, n_base);
            }
        }
    }
    const bool c_cap = sphere && d_cap < kContactThreshold, c_base = sphere && d_base < kContactThreshold;
    {
        double *m = scratch + SC_STASH_MISC + L.l;
#pragma unroll
        for (int k = 0; k < 3; k++) { m[(SM_CC + k) * GL] = cc[k]; m[(SM_NCAP + k) * GL] = n_cap[k]; m[(SM_NBASE + k) * GL] = n_base[k]; }
        m[SM_DCAP * GL] = d_cap; m[SM_DBASE * GL] = d_base;
    }
    P.contact_table = gany(sphere && (cc[2] - L.sph(3) - L.table_z() < kContactThreshold)) ? 1 : 0;
    P.contact_button = gany(c_cap) ? 1 : 0;
    const double qd = g.qd * L.jm;
    P.qd = qd;
    tdynamics(L, g, S, qd, W, P.tau);
#pragma unroll
    for (int k = 0; k < NJ; k++) (scratch + SC_STASH_W)[k * GL + L.l] = W[k];
    double qdd = 0.0;
    rdot_step<0, NJ>(qdd, W, P.tau);
    const double qd_new = fma(dt, qdd, qd);
    P.qd_new = qd_new;
    const double bqd_g = e.bqd + dt * kGravityZ;
    P.bqd_g = bqd_g;
    // ---- bank-A rows (impulse space, A = J W J^T): motor row per joint lane, the button's three scalar rows
    const double wb = 1.0 / kCapMass, blim = kLimitMaxImpulse;
    const double bound_bm = motor_on ? kButtonMaxForce * dt : kDefaultMotorImpulse;
    const bool is_bm = L.l == kBM, is_blo = L.l == kBLo, is_bhi = L.l == kBHi;
    const double lerp = L.limit_erp();
    TRows &r = P.r;
    double rhs = 0.0, off = 0.0;
#pragma unroll
    for (int k = 0; k < GL; k++) r.n[k] = 0.0;
    r.lo = 0.0; r.S = 0.0; r.jb = 0.0; r.diag = 0.0;
    // bounds of every bank-A row are known on every lane without communication: motor row k has lo = -bound_k, S = 2 bound_k (the
    // lane table), the button motor -+bound_bm, the button stops [0, blim]
    auto S_of = [&](int k) -> double { return k < NJ ? 2.0 * tab[LT_BOUND * GL + k] : k == kBM ? 2.0 * bound_bm : k < GL - 1 ? blim : 0.0; };
    auto lo_of = [&](int k) -> double { return k < NJ ? -tab[LT_BOUND * GL + k] : k == kBM ? -bound_bm : 0.0; };
    // unscaled coupling of the own bank-A row to bank-A row k
    auto a_of = [&](int k) -> double {
        if (k < NJ) return L.jnt ? W[k] * (1.0 - L.e(k)) : 0.0;
        if (k == kBM) return is_blo ? wb : is_bhi ? -wb : 0.0;
        if (k == kBLo) return is_bm ? wb : is_bhi ? -wb : 0.0;
        if (k == kBHi) return (is_bm || is_blo) ? -wb : 0.0;
        return 0.0;
    };
    if (L.jnt) {
#pragma unroll
        for (int k = 0; k < NJ; k++) r.diag = fma(L.e(k), W[k], r.diag);
        r.lo = -L.bound(); r.S = 2.0 * L.bound();       // (rhs = target - qd_new: tphysics_post2, behind the IK)
    } else if (is_bm) {
        rhs = (motor_on ? kButtonKp * (kButtonTarget - e.bq) * inv_dt : 0.0) - bqd_g;
        r.lo = -bound_bm; r.S = 2.0 * bound_bm; r.jb = 1.0; r.diag = wb;
    } else if (is_blo) {
        const double pen = e.bq - kGliderLower;
        rhs = ((pen > 0 ? -pen * inv_dt : 0.0) - bqd_g) + (pen > 0 ? 0.0 : -pen * lerp * inv_dt);
        r.S = blim; r.jb = 1.0; r.diag = wb;
    } else if (is_bhi) {
        const double pen = kGliderUpper - e.bq;
        rhs = ((pen > 0 ? -pen * inv_dt : 0.0) + bqd_g) + (pen > 0 ? 0.0 : -pen * lerp * inv_dt);
        r.S = blim; r.jb = -1.0; r.diag = wb;
    }
#pragma unroll
    for (int k = 0; k < kNArows; k++) off = fma(a_of(k), lo_of(k), off);
    P.bound_bm = bound_bm; P.rhs_b = rhs; P.off = off;
    // ---- generic rows: joint-limit candidates of the own joint, contact candidates of the own sphere
    const bool has_lim = L.jnt && L.jlo() <= L.jhi();
    const double pen_lo = g.q - L.jlo(), pen_hi = L.jhi() - g.q;
    const bool lim_lo = has_lim && pen_lo <= kLimitActivationVel * dt, lim_hi = has_lim && pen_hi <= kLimitActivationVel * dt;
    { scratch[SC_STASH_MISC + SM_PENLO * GL + L.l] = pen_lo; scratch[SC_STASH_MISC + SM_PENHI * GL + L.l] = pen_hi; }
    const bool any_generic = wany(lim_lo || lim_hi || c_cap || c_base);
    P.any_generic = any_generic;
    // ---- scale the bank-A rows to u in [0, 1]:  x_r = cs_r + sum_k n_rk u_k  (cs_r: tphysics_post2)
    {
        const bool live = r.S > 0.0 && r.diag > 0.0;
        const double inv = live ? rcp(r.diag * r.S) : 0.0;
        P.live = live; P.inv = inv;
#pragma unroll
        for (int k = 0; k < kNArows; k++) r.n[k] = -(a_of(k) * S_of(k)) * inv;
        r.acc0 = 0.0;
        if (L.jnt) {
#pragma unroll
            for (int k = 0; k < NJ; k++) r.acc0 = fma(r.n[k] * (k > L.l ? 1.0 : 0.0), 0.5, r.acc0);
        }
    }
    if (any_generic) {
        GenIn in;
        in.tab = tab; in.scratch = scratch; in.r = r; in.qd_new = qd_new; in.bqd = bqd_g; in.bound_bm = bound_bm;
        in.r2.cs = 0.0; in.r2.n[0] = 0.0; in.r2.n[1] = 0.0; in.r2.n[2] = 0.0; in.r2.lo = 0.0; in.r2.S = 0.0; in.r2.jb = 0.0;
        in.bqd2 = 0.0; in.rb = nullptr; in.park = nullptr; in.g = &g; in.e = &e;
        (void)general_path<1, 0, 0, DET, 1>(in, &P.ctx);
    }
}
template <int DET>
SRL_G void tphysics_post2(Env &e, GState &g, const double *tab, const Cfg &cfg, double *scratch, const double motor[3], bool joint_mode, double jt_own,
                          double finger_angle, PreStep &P) {
    const double dt = kDt, inv_dt = 1.0 / kDt;
    const TL L = lane_view(tab);
    const double *S = P.S;
    e.contact_table = P.contact_table; e.contact_button = P.contact_button;
    // ---- IK target accumulate + clip (kuka.py:134-139), one damped-least-squares step on the arm block (kuka.py:144-156)
    double qdes = L.arm ? jt_own : L.tsel() * finger_angle;
    if (!joint_mode) {
        const int b = (cfg.random_target || cfg.two) ? 0 : 1;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            double v = e.ee[k] + motor[k];
            v = v < kEeBox[b][0][k] ? kEeBox[b][0][k] : v;
            v = v > kEeBox[b][1][k] ? kEeBox[b][1][k] : v;
            e.ee[k] = v;
        }
        double Rt[9], pt[3], ee[3], dS[6], J[6];
#pragma unroll
        for (int k = 0; k < 9; k++) Rt[k] = bcast<NA - 1>(g.R[k]);
#pragma unroll
        for (int k = 0; k < 3; k++) pt[k] = bcast<NA - 1>(g.p[k]);
        const double eept[3] = {L.eept(0), L.eept(1), L.eept(2)};
        frame_point(Rt, pt, eept, ee);
        {
            double d[3], Sa[3];
#pragma unroll
            for (int k = 0; k < 3; k++) { d[k] = ee[k] - g.p[k]; Sa[k] = S[k] * L.am; }
            cross3(Sa, d, J);
#pragma unroll
            for (int k = 0; k < 3; k++) J[3 + k] = Sa[k];
        }
#pragma unroll
        for (int k = 0; k < 3; k++) dS[k] = e.ee[k] - ee[k];
        {   // orientation error towards quat(euler(0, -pi, 0)), replicated (same construction as kuka_group.hpp)
            const double *R = Rt;
            double qx, qy, qz, qw;
            const double m00 = R[0], m01 = R[3], m02 = R[6], m10 = R[1], m11 = R[4], m12 = R[7], m20 = R[2], m21 = R[5], m22 = R[8];
            const double tr = m00 + m11 + m22;
            if (tr > 0) { double s = sqrt(tr + 1.0) * 2; qw = 0.25 * s; qx = (m21 - m12) / s; qy = (m02 - m20) / s; qz = (m10 - m01) / s; }
            else if (m00 > m11 && m00 > m22) { double s = sqrt(1.0 + m00 - m11 - m22) * 2; qw = (m21 - m12) / s; qx = 0.25 * s; qy = (m01 + m10) / s; qz = (m02 + m20) / s; }
            else if (m11 > m22) { double s = sqrt(1.0 + m11 - m00 - m22) * 2; qw = (m02 - m20) / s; qx = (m01 + m10) / s; qy = 0.25 * s; qz = (m12 + m21) / s; }
            else { double s = sqrt(1.0 + m22 - m00 - m11) * 2; qw = (m10 - m01) / s; qx = (m02 + m20) / s; qy = (m12 + m21) / s; qz = 0.25 * s; }
            const double tx = 0.0, ty = -1.0, tz = 0.0, tw = 6.123233995736766e-17;
            const double ix = -qx, iy = -qy, iz = -qz, iw = qw;
            const double dw = tw * iw - tx * ix - ty * iy - tz * iz;
            const double dx = tw * ix + tx * iw + ty * iz - tz * iy;
            const double dy = tw * iy - tx * iz + ty * iw + tz * ix;
            const double dz = tw * iz + tx * iy - ty * ix + tz * iw;
            const double sv = sqrt(dx * dx + dy * dy + dz * dz);
            double angle = 2.0 * atan2(sv, dw), ax, ay, az;
            if (sv * sv < 10.0 * 2.2204460492503131e-16) { ax = 1; ay = 0; az = 0; }
            else { ax = dx / sv; ay = dy / sv; az = dz / sv; }
            if (angle > kPi) angle -= 2 * kPi;
            dS[3] = angle * ax; dS[4] = angle * ay; dS[5] = angle * az;
        }
        double A[NA], bb = 0.0;
        dot6_step<0, NA>(J, J, A);
        const double damping = cfg.two ? kIkDampingDefault : kIkDamping;
#pragma unroll
        for (int k = 0; k < NA; k++) A[k] = fma(damping, L.e(k), A[k]);
#pragma unroll
        for (int c = 0; c < 6; c++) bb = fma(J[c], dS[c], bb);
        double det = 1.0;
        gj_step<0, NA, false>(L, A, bb, &det);
        if (det < kIkCrossDet) e.ikx |= 1;      // sticky until the episode's reset (kuka_core.hpp kIkCrossDet)
        bb *= L.am;
        double all[NA], maxabs = 0.0;
        ball_step<0, NA>(bb, all);
#pragma unroll
        for (int k = 0; k < NA; k++) maxabs = fmax(maxabs, fabs(all[k]));
        if (L.arm) qdes = g.q + bb;
        if (wany(maxabs > kIkMaxAngle)) {
            const double scale = kIkMaxAngle / maxabs;
            if (maxabs > kIkMaxAngle && L.arm) qdes = g.q + bb * scale;
        }
    }
    double target = L.kp() * (qdes - g.q) * inv_dt;
    target = target > L.maxvel() ? L.maxvel() : target;
    target = target < -L.maxvel() ? -L.maxvel() : target;
    const double qd_new = P.qd_new, wb = 1.0 / kCapMass;
    e.bqd = P.bqd_g;
    TRows &r = P.r;
    const bool is_bm = L.l == kBM;
    {
        const double rhs = L.jnt ? target - qd_new : P.rhs_b, off = P.off, inv = P.inv;
        const bool live = P.live;
        r.cs = live ? (rhs - off) * inv + ((L.jnt || is_bm) ? 0.5 : 0.0) : 0.0;
    }
    static_assert(DET == 0, "the early setup exists for the default solver details (the configuration-specialised kernel)");
    double u, acc_b = 0.0, dvb_b = 0.0;
    if (!P.any_generic) u = sweeps_free<1>(r);
    else {
        GenIn in;
        in.tab = tab; in.scratch = scratch; in.r = r; in.qd_new = qd_new; in.bqd = e.bqd; in.bound_bm = P.bound_bm;
        in.r2.cs = 0.0; in.r2.n[0] = 0.0; in.r2.n[1] = 0.0; in.r2.n[2] = 0.0; in.r2.lo = 0.0; in.r2.S = 0.0; in.r2.jb = 0.0;
        in.bqd2 = 0.0; in.rb = nullptr; in.park = nullptr; in.g = &g; in.e = &e;
        const GenOut out = general_path<1, 0, 0, DET, 2>(in, &P.ctx);
        u = out.u; acc_b = out.acc_b; dvb_b = out.dvb_b;
    }
    const double lam = r.lo + r.S * u;
    SRL_GDBG(5, lane_id(), lam);
    // ---- velocity change: joint lane i gets sum_r a_ir lambda_r, the glider sum_r jb_r lambda_r / m
    double dv = 0.0, dvb = 0.0;
    {
        const double v = r.S > 0.0 ? lam / r.S : 0.0, pb = r.jb * lam * wb;
        double acc = 0.0;
#define SRL_ACC(K) fmac_bcast<K>(acc, v, r.n[K]);
        SRL_ACC(0) SRL_ACC(1) SRL_ACC(2) SRL_ACC(3) SRL_ACC(4) SRL_ACC(5) SRL_ACC(6) SRL_ACC(7) SRL_ACC(8) SRL_ACC(9) SRL_ACC(10) SRL_ACC(11)
#undef SRL_ACC
        dvb = bcast<kBM>(pb) + bcast<kBLo>(pb) + bcast<kBHi>(pb);
        acc += acc_b; dvb += dvb_b;
        dv = r.diag * (lam - r.S * acc);
    }
    // ---- semi-implicit Euler, refresh sin/cos, frames and the gripper position (a fresh copy of the lane constants: the one loaded
    //      at the top of the step is not kept live across the solver loop)
    if (lane_id() < NJ) { g.qd = qd_new + dv; g.q += dt * g.qd; }
    e.bqd += dvb;
    e.bq += dt * e.bqd;
    const TL L3 = lane_view(tab);
    trefresh(L3, g, e);
}

// ------------------------------------------------------------------ env level (mirrors kuka_group.hpp / kuka_env.hpp)
// packed start state: q12 qd12 sq12 cq12 ee3 bq bqd grip3
SRL_G void tunpack_start(Env &e, GState &g, const double *o) {
    const int l = lane_id();
    if (l < NJ) { g.q = o[l]; g.qd = o[NJ + l]; g.sq = o[2 * NJ + l]; g.cq = o[3 * NJ + l]; }
    else { g.q = 0.0; g.qd = 0.0; g.sq = 0.0; g.cq = 1.0; }
#pragma unroll
    for (int k = 0; k < 3; k++) { e.ee[k] = o[4 * NJ + k]; e.grip[k] = o[4 * NJ + 5 + k]; }
    e.bq = o[4 * NJ + 3]; e.bqd = o[4 * NJ + 4];
}
SRL_G void tpack_start(const Env &e, const GState &g, double *o) {
    const int l = lane_id();
    if (l < NJ) { o[l] = g.q; o[NJ + l] = g.qd; o[2 * NJ + l] = g.sq; o[3 * NJ + l] = g.cq; }
    if (l == 0) {
#pragma unroll
        for (int k = 0; k < 3; k++) { o[4 * NJ + k] = e.ee[k]; o[4 * NJ + 5 + k] = e.grip[k]; }
        o[4 * NJ + 3] = e.bq; o[4 * NJ + 4] = e.bqd;
    }
}
// state right after loadSDF / resetJointState (kuka.py:56-73): all joints at joint_positions, IK target at its initial value
SRL_G void tinitial(Env &e, GState &g, const double *tab) {
    const TL L = lane_view(tab);
    g.q = L.jnt ? L.q0() : 0.0; g.qd = 0.0; g.sq = 0.0; g.cq = 1.0;
#pragma unroll
    for (int k = 0; k < 3; k++) { e.ee[k] = kEeInit[k]; e.bpos[k] = 0.0; }
    e.bq = 0.0; e.bqd = 0.0; e.bx = kButtonX; e.by = kButtonY; e.bz = L.base_z(); e.bspeed = 0.0;
    e.motor_on = 0; e.contact_button = 0; e.contact_table = 0; e.counter = 0; e.n_contacts = 0; e.n_outside = 0; e.terminated = 0; e.ikx = 0;
    e.b2q = 0.0; e.b2qd = 0.0; e.b2x = kButtonX; e.b2y = kButton2Y2B; e.contact_body1 = 0; e.contact_body2 = 0; e.goal_id = 0; e.n_contacts2 = 0;
    trefresh(L, g, e);
}

// KukaButtonGymEnv.reset for one lane group: the reference's draws in its order (reset_draw), then the episode's start state.
// START = 0: from the start-state table (Cartesian action modes: the 500 settle steps are RNG- and contact-free and each of the
// five init actions is one of 6 (2) noise-free moves, so an episode starts from one of 6^5 (2^5) states, integrated once per
// handle); START = 1: joint-space actions — the five init actions are integrated here from the settled state; START = 2: Cartesian
// modes without a table (the CPU harness): the same five moves integrated from the settled state.
struct RbResetHook {          // reset_draw's view of the distractor candidates: lane k keeps candidate k
    RBody *B; int l;
    SRL_G void operator()(int i, double ox, double oy, bool keep) const { if (B && i == l) { B->ox = ox; B->oy = oy; B->on = keep; } }
};
template <int START, int NB = 1, int RB = 0, int OCC = 0, class R>
SRL_G void tenv_reset(Env &e, GState &g, const double *tab, const Cfg &cfg, double *scratch, R &rng, const double *starts, const double *settled,
                      double *objs, int64_t objs_stride, RBody *rb = nullptr, double *park = nullptr) {
#pragma clang fp contract(off)
    const TL L = lane_view(tab);
    ResetDraw d;
    RbResetHook hook; hook.B = RB ? rb : nullptr; hook.l = L.l;
    reset_draw<NB>(cfg, L.l == 0 ? objs : nullptr, objs_stride, rng, d, hook);
    if constexpr (RB) {
        // the bodies at rest on the table (the reference drops them before its 500 settle steps: oracle/kuka_oracle.c, free-body section)
        RBody &B = *rb;
        if (L.l < 10) { B.type = rb_type_of(B.ox, B.oy); B.x[0] = B.ox; B.x[1] = B.oy; }
        else if (L.l == 10) { B.type = 3; B.on = true; B.x[0] = 0.25; B.x[1] = -0.2; B.ox = 0.0; B.oy = 0.0; }
        else { B.type = 2; B.on = false; B.x[0] = 0.0; B.x[1] = 0.0; B.ox = 0.0; B.oy = 0.0; }
        B.x[2] = L.table_z() + rb_height(B.type);
        B.v[0] = 0.0; B.v[1] = 0.0; B.v[2] = 0.0;
    }
    e.motor_on = 0; e.contact_button = 0; e.contact_table = 0;
    tunpack_start(e, g, START ? settled : starts + (int64_t)d.idx * kTreeStartDoubles);
    e.bx = d.bx; e.by = d.by; e.bz = L.base_z();
    if constexpr (NB == 2) { e.b2x = d.b2x; e.b2y = d.b2y; e.b2q = e.bq; e.b2qd = e.bqd; }    // same urdf, same free steps
    tfk(L, g);
    if constexpr (START == 1) {
        const double motor[3] = {0, 0, 0};
        for (int k = 0; k < kNInitActions; k++) {
            const double jt = L.q0() + kDeltaTheta * d.g[k];
            tphysics_step<NB, RB, OCC>(e, g, tab, cfg, scratch, motor, true, jt, 0.0, rb, park);
        }
    } else if constexpr (START == 2) {
        const int base = cfg.is_discrete ? 6 : 2;
        int rem = d.idx;
        double motor[3];
        for (int k = 0; k < kNInitActions; k++) {
            init_action_motor(cfg, rem % base, motor);
            tphysics_step<NB, RB, OCC>(e, g, tab, cfg, scratch, motor, false, L.q0(), 0.0, rb, park);
            rem /= base;
        }
    }
    reset_finish<NB>(e, d, L.base_z());
}

// KukaButtonGymEnv.step + step2 for one lane group.  ca3: the Cartesian action (replicated), ca_own: the own arm joint's action.
// finger_angle = 0.0 (kuka_button_gym_env.py:312,335: "Close the gripper"; joints mode appends [0, 0]).
// EARLY = 1 (persistent kernels): `pre` holds the action-independent half of the FIRST physics step (tphysics_pre on the state this call
// starts from); EARLY = 2 (the configuration-specialised persistent kernel): `pre2` holds everything in front of the action (tphysics_pre2)
template <int NB = 1, int RB = 0, int OCC = 0, int DET = -1, int EARLY = 0, class R>
SRL_G double tenv_step(Env &e, GState &g, const double *tab, const Cfg &cfg, double *scratch, R &rng, int action, const float *ca3, float ca_own, bool *done,
                       RBody *rb = nullptr, double *park = nullptr, const PreDyn *pre = nullptr, PreStep *pre2 = nullptr) {
    if constexpr (RB) {
        // kuka_rand_button_gym_env.py:111-123: at env step 10 the ball is kicked (applyExternalForce: it acts on the next stepSimulation)
        const double kx = shfl(rb->ox, 9), ky = shfl(rb->oy, 9);
        if (e.counter == kRbKickStep && lane_id() == 10) {
#pragma clang fp contract(off)
            double f[3];
            rb_kick_force(kx, ky, f);
#pragma unroll
            for (int k = 0; k < 3; k++) rb->v[k] += f[k] * kDt / kRbMass;
        }
    }
    StepCmd c;
    step_command(e, cfg, rng, action, ca3, c);
    const double jt = joint_target(c, ca_own, tab[LT_Q0 * GL + lane_id()]);
    SRL_TSTAMP(21);                         // noise draw + action mapping (step_command)
    for (int rep = 0; rep < cfg.action_repeat; rep++) {
        if constexpr (EARLY == 2) {
            if (rep == 0 && !wany(e.motor_on != 1)) tphysics_post2<DET>(e, g, tab, cfg, scratch, c.motor, c.joint_mode, jt, 0.0, *pre2);
            else tphysics_step<NB, RB, OCC, DET>(e, g, tab, cfg, scratch, c.motor, c.joint_mode, jt, 0.0, rb, park);
        } else if constexpr (EARLY) {
            if (rep == 0) tphysics_step<NB, RB, OCC, DET, 1>(e, g, tab, cfg, scratch, c.motor, c.joint_mode, jt, 0.0, rb, park, pre);
            else tphysics_step<NB, RB, OCC, DET>(e, g, tab, cfg, scratch, c.motor, c.joint_mode, jt, 0.0, rb, park);
        } else {
            tphysics_step<NB, RB, OCC, DET>(e, g, tab, cfg, scratch, c.motor, c.joint_mode, jt, 0.0, rb, park);
        }
        if (termination(e, cfg)) break;
        e.counter += 1;
    }
    e.ikx += (e.ikx & 1) << 1;               // env-steps taken under the conditioning flag
    const double reward = NB == 2 ? reward_two(e, cfg) : reward_fn(e, cfg);
    *done = termination(e, cfg);
    SRL_TSTAMP(10);                         // counters, reward (one float64 sqrt), termination
    return reward;
}

}  // namespace tree
}  // namespace kuka
}  // namespace srl
