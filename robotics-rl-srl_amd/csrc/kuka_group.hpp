// kuka_group.hpp — lane-GROUP formulation of the KukaButtonGymEnv physics step for gfx950: 16 lanes (one DPP row of a
// wavefront) integrate ONE env, so a 4096-env batch launches 1024 wavefronts (one per SIMD of the MI355X) instead of the
// 64 of the lane-per-env kernel (kuka_core.hpp), and the sequential part of a step shrinks from ~17 000 to ~5 000
// instructions per wavefront.  Same model, same row semantics, same env wrapper (kuka_env.hpp) as the lane-per-env kernel
// — only the decomposition differs:
//
//   lane l < 7 owns joint / link l (q, qd, sin/cos, world frame, spatial axis, inertia, row l of M^-1, arm-motor row l);
//   lanes 8, 9, 10 own the button's scalar rows (motor, lower stop, upper stop); lanes 7, 11..15 the <= 6 generic rows
//   (arm joint limits, gripper-sphere contacts) — one constraint row per lane.
//
//   * kinematics: the seven joint transforms are composed by a 3-level parallel prefix (row_shr DPP), so every lane
//     gets its own link's world frame in 3 compositions instead of 7;
//   * dynamics in WORLD coordinates about the world origin, where the recursions of RNEA / CRBA are plain prefix /
//     suffix sums over the chain: link velocities and bias accelerations (prefix), link forces and composite inertias
//     (suffix) are masked row-broadcast FMAs (v_fmac_f64_dpp row_newbcast — the only DPP control gfx950 has for 64-bit
//     operands); M (CRBA) is inverted in place by a lane-parallel Gauss-Jordan sweep (row i on lane i);
//   * IK: Jacobian column per lane, J^T J by broadcast FMAs, the 7x7 SPD solve by the same Gauss-Jordan scheme;
//   * projected Gauss-Seidel in impulse space with one row per lane, every row rescaled to u = (lambda - lo) / (hi - lo)
//     in [0, 1] so that the projection is the hardware clamp modifier of the add that forms the row's residual:
//     a row update is  t = clamp01(cs + acc);  acc -= e_prev * acc;  acc += n_j * bcast_j(t)  — three instructions,
//     the "wavefront-level reduction for contact resolution" of the north star done as broadcast-accumulate.
//
// Numerics: float64 like the reference; the association order of sums differs from kuka_core.hpp / the oracle
// (parallel prefix instead of chain walks, Gauss-Jordan instead of LDL^T / ABA), so results agree to ~1e-11, not bit
// for bit — the north-star bar is 1e-4 on joints and bit-exact discrete flags, which the parity tests check.
//
// The same source is compiled for the host by the CPU-side parity harness (csrc/kuka_hostcheck.cpp, tests only): there
// the 16 lanes of a group are 16 cooperatively scheduled fibers and the cross-lane primitives below go through a small
// exchange runtime.
#pragma once
#include "kuka_env.hpp"

#ifndef SRL_GDBG          // host-side instrumentation of the test harness only (kuka_hostcheck.cpp)
#define SRL_GDBG(tag, idx, val)
#endif

namespace srl {
namespace kuka {
namespace grp {

constexpr int GL = 16;                 // lanes per env = one DPP row
constexpr int kBM = 8, kBLo = 9, kBHi = 10;                       // lanes of the button's scalar rows
constexpr int kGenLane[kMaxGenRows] = {7, 11, 12, 13, 14, 15};    // lanes of the generic rows, in creation order
// per-group scratch for the (rare) generic-row setup: [row][12] definitions + [row][7] W J
constexpr int kRowDef = 12, kScratchDoubles = kMaxGenRows * (kRowDef + ND);

// ------------------------------------------------------------------ cross-lane primitives
// Device bodies: DPP / ballot instructions.  Host bodies (the host pass of hipcc never calls them; the parity harness
// csrc/kuka_hostcheck.cpp does): the calling fiber's lane, a lockstep value exchange, a group vote.
#define SRL_G __host__ __device__ __forceinline__
int host_lane();
double host_exchange(double x, int src);
uint32_t host_ballot(bool p);
#if defined(__HIP_DEVICE_COMPILE__)
#define SRL_G_DEVICE 1
#else
#define SRL_G_DEVICE 0
#endif

constexpr int kRowsPerWave = 4;       // 16-lane rows (envs) per wavefront
SRL_G int row_id() {                  // the own row inside the wavefront (the fiber harness runs one env per group: row 0)
#if SRL_G_DEVICE
    return (int)((threadIdx.x & 63) >> 4);
#else
    return 0;
#endif
}
SRL_G int lane_id() {
#if SRL_G_DEVICE
    return (int)(threadIdx.x & (GL - 1));
#else
    return host_lane();
#endif
}
template <int J> SRL_G double bcast(double x) {                                   // row_newbcast:J
#if SRL_G_DEVICE
    return __builtin_amdgcn_update_dpp(x, x, 0x150 + J, 0xf, 0xf, false);
#else
    return host_exchange(x, J);
#endif
}
template <int D> SRL_G double shr(double x, double fill) {                        // row_shr:D, `fill` shifted in
#if SRL_G_DEVICE
    const long long v = __double_as_longlong(x), f = __double_as_longlong(fill);
    const int lo = __builtin_amdgcn_update_dpp((int)f, (int)v, 0x110 + D, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp((int)(f >> 32), (int)(v >> 32), 0x110 + D, 0xf, 0xf, false);
    return __longlong_as_double(((long long)(unsigned)hi << 32) | (unsigned)lo);
#else
    const int l = host_lane();
    const double v = host_exchange(x, l >= D ? l - D : l);
    return l >= D ? v : fill;
#endif
}
SRL_G uint32_t ballot(bool p) {
#if SRL_G_DEVICE
    return (uint32_t)(__ballot(p) >> (threadIdx.x & 48)) & 0xffffu;
#else
    return host_ballot(p);
#endif
}
SRL_G bool gany(bool p) { return ballot(p) != 0; }
SRL_G double shfl(double x, int src) {                                            // value of lane `src` of the row (src varies per row)
#if SRL_G_DEVICE
    return __shfl(x, src, GL);
#else
    return host_exchange(x, src);
#endif
}
SRL_G bool wany(bool p) {
#if SRL_G_DEVICE
    return __any(p);
#else
    return host_ballot(p) != 0;
#endif
}
SRL_G void sync_scratch() {
#if SRL_G_DEVICE
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
#else
    (void)host_ballot(false);
#endif
}
SRL_G double rcp(double x) {          // ~1 ulp reciprocal: v_rcp_f64 + two Newton steps
#if SRL_G_DEVICE
    double r = __builtin_amdgcn_rcp(x);
    double e = fma(-x, r, 1.0); r = fma(r, e, r);
    e = fma(-x, r, 1.0); r = fma(r, e, r);
    return r;
#else
    return 1.0 / x;
#endif
}
// acc += w * bcast<J>(x) in one v_fmac_f64_dpp; `s_nop 1` = the two wait states a DPP read needs after a VALU write of x
template <int J> SRL_G void fmac_bcast(double &acc, double x, double w) {
#if SRL_G_DEVICE
    // (not volatile: a pure function of its operands, so independent chains of these can be interleaved by the scheduler —
    //  a dependent f64 instruction issues every ~9 cycles on gfx950, an independent one every ~5)
    asm("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(w), "n"(J));
#else
    acc = fma(host_exchange(x, J), w, acc);
#endif
}
SRL_G double clamp01(double s) { return s < 0.0 ? 0.0 : (s > 1.0 ? 1.0 : s); }
// One projected Gauss-Seidel row update (file header), rows in [0, 1] units:
//     t = clamp01(cs + acc);  acc -= ep * acc;  acc += n * bcast<J>(t)        -> returns t (only lane J's t means something)
// i.e. v_add_f64 ... clamp / v_fma_f64 / v_fmac_f64_dpp.  The reset FMA and one s_nop are the two wait states a DPP read of
// t needs after the VALU write of t.  Each row is ONE asm statement so that nothing else is scheduled into it.
template <int J> SRL_G double pgs_row(double &acc, double cs, double n, double ep) {
#if SRL_G_DEVICE
    double t;
    asm volatile("v_add_f64 %1, %2, %0 clamp\n\tv_fma_f64 %0, -%4, %0, %0\n\ts_nop 0\n\t"
                 "v_fmac_f64_dpp %0, %1, %3 row_newbcast:%5 row_mask:0xf bank_mask:0xf"
                 : "+v"(acc), "=&v"(t) : "v"(cs), "v"(n), "v"(ep), "n"(J));
    return t;
#else
    const double t = clamp01(cs + acc);
    acc = fma(-ep, acc, acc);
    acc = fma(host_exchange(t, J), n, acc);
    return t;
#endif
}
// two mutually decoupled rows (lanes J and J2) updated by the same instructions
template <int J, int J2> SRL_G double pgs_row2(double &acc, double cs, double n, double n2, double ep) {
#if SRL_G_DEVICE
    double t;
    asm volatile("v_add_f64 %1, %2, %0 clamp\n\tv_fma_f64 %0, -%5, %0, %0\n\ts_nop 0\n\t"
                 "v_fmac_f64_dpp %0, %1, %3 row_newbcast:%6 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %1, %4 row_newbcast:%7 row_mask:0xf bank_mask:0xf"
                 : "+v"(acc), "=&v"(t) : "v"(cs), "v"(n), "v"(n2), "v"(ep), "n"(J), "n"(J2));
    return t;
#else
    const double t = clamp01(cs + acc);
    acc = fma(-ep, acc, acc);
    const double a = host_exchange(t, J), b = host_exchange(t, J2);
    acc = fma(a, n, acc);
    acc = fma(b, n2, acc);
    return t;
#endif
}

// ------------------------------------------------------------------ per-lane constants
// Kept in registers for the whole rollout, so only what every step needs; chain masks are rebuilt per step (Masks).
struct Lane {
    int l;
    bool arm;                 // l < 7: owns a joint / link
    double am;                // 1.0 on arm lanes
    double e[ND];             // e[j] = (l == j), j < 7
    double mass, mcomp;       // link mass, composite mass of links l..6
    double com[3], in[3];     // centre of mass and principal inertia in the link frame
    // joint frame in the parent link's frame.  Baked model (CM = false), branch-free: columns x' = (sx c, f0 s, (1-f0) s),
    // y' = (-sx s, f0 c, (1-f0) c), z' = (0, zy, f0) (kFix 0: Rz(q); 1: rpy (pi/2,0,pi) Rz(q); 2: rpy (pi/2,0,0) Rz(q); identity
    // off the arm).  Runtime table (CM = true): F = the fixed rotation's columns, x' = c Fx + s Fy, y' = c Fy - s Fx, z' = Fz.
    double sx, f0, zy, t[3];
    double F[9];
    double jlo, jhi, q0;      // joint limits, kJointPositions[l]
    double sph[4];            // l < 6: gripper sphere l (centre in the link-7 frame, radius)
    // replicated scalars of the model (compile-time constants unless a table is installed)
    double damping, eept[3], grpt[3], table_z, base_z;
};

// URDF rpy -> rotation (Rz(yaw) Ry(pitch) Rx(roll)), columns x y z; entries within 1e-12 of 0 / +-1 are snapped (the iiwa
// chain is made of quarter turns, which pi/2 in floating point only approximates)
SRL_G void rpy_columns(const double rpy[3], double F[9]) {
    const double sr = sin(rpy[0]), cr = cos(rpy[0]), sp = sin(rpy[1]), cp = cos(rpy[1]), sy = sin(rpy[2]), cy = cos(rpy[2]);
    F[0] = cy * cp; F[1] = sy * cp; F[2] = -sp;
    F[3] = cy * sp * sr - sy * cr; F[4] = sy * sp * sr + cy * cr; F[5] = cp * sr;
    F[6] = cy * sp * cr + sy * sr; F[7] = sy * sp * cr - cy * sr; F[8] = cp * cr;
#pragma unroll
    for (int k = 0; k < 9; k++) {
        if (fabs(F[k]) < 1e-12) F[k] = 0.0;
        else if (fabs(F[k] - 1.0) < 1e-12) F[k] = 1.0;
        else if (fabs(F[k] + 1.0) < 1e-12) F[k] = -1.0;
    }
}

// CM = false: the baked model (constants of kuka_core.hpp).  CM = true: the runtime table `m` (device or host memory).
template <bool CM>
SRL_G void lane_init(Lane &L, const Model *m) {
    const int l = lane_id();
    L.l = l; L.arm = l < ND; L.am = L.arm ? 1.0 : 0.0;
#pragma unroll
    for (int j = 0; j < ND; j++) L.e[j] = l == j ? 1.0 : 0.0;
    const int i = L.arm ? l : 0;
    const int s = l < kNSphere ? l : 0;
    L.q0 = kJointPositions[i];
    L.sx = 1.0; L.f0 = 1.0; L.zy = 0.0;
#pragma unroll
    for (int k = 0; k < 9; k++) L.F[k] = (k % 4 == 0) ? 1.0 : 0.0;
    if constexpr (!CM) {
        double mc = 0.0;
#pragma unroll
        for (int k = 0; k < ND; k++) mc += k >= l ? kMass[k] : 0.0;
        L.mass = L.arm ? kMass[i] : 0.0; L.mcomp = L.arm ? mc : 0.0;
#pragma unroll
        for (int k = 0; k < 3; k++) { L.com[k] = kCom[i][k]; L.in[k] = L.arm ? kInertia[i][k] : 0.0; }
        const int fix = L.arm ? kFix[i] : 0, axis = kTransAxis[i];
        const double len = L.arm ? kTransLen[i] : 0.0;
        L.sx = fix == 1 ? -1.0 : 1.0; L.f0 = fix == 0 ? 1.0 : 0.0; L.zy = fix == 1 ? 1.0 : fix == 2 ? -1.0 : 0.0;
        L.t[0] = l == 0 ? kBasePos[0] : 0.0; L.t[1] = (l == 0 ? kBasePos[1] : 0.0) + (axis == 1 ? len : 0.0);
        L.t[2] = (l == 0 ? kBasePos[2] : 0.0) + (axis == 2 ? len : 0.0);
        L.jlo = kJointLower[i]; L.jhi = kJointUpper[i];
#pragma unroll
        for (int k = 0; k < 4; k++) L.sph[k] = kSphere[s][k];
        L.damping = kJointDamping; L.table_z = kTableTopZ; L.base_z = kButtonBaseZ;
#pragma unroll
        for (int k = 0; k < 3; k++) { L.eept[k] = kEePoint[k]; L.grpt[k] = kGripperPoint[k]; }
    } else {
        double mc = 0.0;
        for (int k = 0; k < ND; k++) mc += k >= l ? m->mass[k] : 0.0;
        L.mass = L.arm ? m->mass[i] : 0.0; L.mcomp = L.arm ? mc : 0.0;
        for (int k = 0; k < 3; k++) {
            L.com[k] = m->com[i][k]; L.in[k] = L.arm ? m->inertia[i][k] : 0.0;
            L.t[k] = (L.arm ? m->joint_xyz[i][k] : 0.0) + (l == 0 ? kBasePos[k] : 0.0);
            L.eept[k] = m->ee_point[k]; L.grpt[k] = m->gripper_point[k];
        }
        if (L.arm) rpy_columns(m->joint_rpy[i], L.F);
        L.jlo = m->joint_lower[i]; L.jhi = m->joint_upper[i];
        for (int k = 0; k < 4; k++) L.sph[k] = m->sphere[s][k];
        L.damping = m->joint_damping; L.table_z = m->table_top_z; L.base_z = m->button_base_z;
    }
}

// prefix / suffix masks over the chain, rebuilt inside every step from an opaque copy of the lane index (kept out of the
// rollout loop's live registers on purpose)
struct Masks { double le[ND], ge[ND]; };     // le[k] = (k <= l);  ge[k] = (k >= l) on arm lanes, 0 elsewhere
SRL_G void make_masks(int l, Masks &m) {
#if SRL_G_DEVICE
    asm volatile("" : "+v"(l));
#endif
#pragma unroll
    for (int k = 0; k < ND; k++) { m.le[k] = k <= l ? 1.0 : 0.0; m.ge[k] = (k >= l && l < ND) ? 1.0 : 0.0; }
}

// ------------------------------------------------------------------ per-lane dynamic state kept across steps
struct GState {
    double q, qd, sq, cq;      // own joint (arm lanes; 0 / 0 / 0 / 1 elsewhere)
    double R[9], p[3];         // world frame of the own link (columns x y z; lanes >= 7: unused)
};

SRL_G void compose(const double Ra[9], const double pa[3], const double Rb[9], const double pb[3], double Ro[9], double po[3]) {
#pragma unroll
    for (int j = 0; j < 3; j++)
#pragma unroll
        for (int k = 0; k < 3; k++) Ro[3 * j + k] = Ra[k] * Rb[3 * j] + Ra[3 + k] * Rb[3 * j + 1] + Ra[6 + k] * Rb[3 * j + 2];
#pragma unroll
    for (int k = 0; k < 3; k++) po[k] = pa[k] + Ra[k] * pb[0] + Ra[3 + k] * pb[1] + Ra[6 + k] * pb[2];
}

// Forward kinematics of the whole chain: local joint transform per lane, inclusive prefix composition over the row.
template <bool CM>
SRL_G void gfk(const Lane &L, GState &g) {
    const double s = g.sq, c = g.cq;                       // lanes off the arm carry s = 0, c = 1: the identity
    double R[9], p[3];
    if constexpr (!CM) {
        const double f1 = 1.0 - L.f0;
        R[0] = L.sx * c; R[1] = L.f0 * s; R[2] = f1 * s;
        R[3] = -(L.sx * s); R[4] = L.f0 * c; R[5] = f1 * c;
        R[6] = 0.0; R[7] = L.zy; R[8] = L.f0;
    } else {
#pragma unroll
        for (int k = 0; k < 3; k++) { R[k] = c * L.F[k] + s * L.F[3 + k]; R[3 + k] = c * L.F[3 + k] - s * L.F[k]; R[6 + k] = L.F[6 + k]; }
    }
    p[0] = L.t[0]; p[1] = L.t[1]; p[2] = L.t[2];
#define SRL_SCAN(D)                                                                              \
    {                                                                                            \
        double Ra[9], pa[3], Ro[9], po[3];                                                       \
        _Pragma("unroll") for (int k = 0; k < 9; k++) Ra[k] = shr<D>(R[k], (k % 4 == 0) ? 1.0 : 0.0); \
        _Pragma("unroll") for (int k = 0; k < 3; k++) pa[k] = shr<D>(p[k], 0.0);                 \
        compose(Ra, pa, R, p, Ro, po);                                                           \
        _Pragma("unroll") for (int k = 0; k < 9; k++) R[k] = Ro[k];                              \
        _Pragma("unroll") for (int k = 0; k < 3; k++) p[k] = po[k];                              \
    }
    SRL_SCAN(1) SRL_SCAN(2) SRL_SCAN(4)
#undef SRL_SCAN
#pragma unroll
    for (int k = 0; k < 9; k++) g.R[k] = R[k];
#pragma unroll
    for (int k = 0; k < 3; k++) g.p[k] = p[k];
}
// world frame of link 7, replicated on the row
SRL_G void tip_frame(const GState &g, double Rt[9], double pt[3]) {
#pragma unroll
    for (int k = 0; k < 9; k++) Rt[k] = bcast<ND - 1>(g.R[k]);
#pragma unroll
    for (int k = 0; k < 3; k++) pt[k] = bcast<ND - 1>(g.p[k]);
}

// sin/cos of the own joint, frames, gripper position (what update_trig_and_gripper() does for the lane-per-env kernel)
template <bool CM>
SRL_G void grefresh(const Lane &L, GState &g, Env &e) {
    if (L.arm) sincos(g.q, &g.sq, &g.cq); else { g.sq = 0.0; g.cq = 1.0; }
    gfk<CM>(L, g);
    double Rt[9], pt[3];
    tip_frame(g, Rt, pt);
    tip_point(Rt, pt, L.grpt, e.grip);
}

// prefix / suffix sums over the chain by masked row broadcasts: out = base + sum_k m[k] * bcast_k(x)
template <int K> SRL_G void masked_sum_step(double &acc, double x, const double m[ND]) {
    fmac_bcast<K>(acc, x, m[K]);
    if constexpr (K + 1 < ND) masked_sum_step<K + 1>(acc, x, m);
}
SRL_G double masked_sum(double x, const double m[ND], double base = 0.0) {
    double acc = base;
    masked_sum_step<0>(acc, x, m);
    return acc;
}

// In-place Gauss-Jordan on the rows of a 7x7 SPD matrix held one row per lane (lanes >= 7 carry zero rows).
// INV = true: A <- A^-1.  INV = false: A x = b, solution left in b (columns <= the pivot are not maintained).
template <int K, bool INV> SRL_G void gj_step(const Lane &L, double A[ND], double &b) {
    const double r = rcp(bcast<K>(A[K]));
    const double g = -((A[K] - L.e[K]) * r);          // lane K: -(1 - 1/pivot): its row ends up scaled by 1/pivot
    if constexpr (INV) {
        A[K] = L.e[K];                                  // the identity's column K takes the place of column K
#pragma unroll
        for (int c = 0; c < ND; c++) fmac_bcast<K>(A[c], A[c], g);
    } else {
#pragma unroll
        for (int c = K + 1; c < ND; c++) fmac_bcast<K>(A[c], A[c], g);
        fmac_bcast<K>(b, b, g);
    }
    if constexpr (K + 1 < ND) gj_step<K + 1, INV>(L, A, b);
}

template <int K> SRL_G void row_dot_step(double &acc, const double row[ND], double x) {
    fmac_bcast<K>(acc, x, row[K]);
    if constexpr (K + 1 < ND) row_dot_step<K + 1>(acc, row, x);
}
// sum_k row[k] * x_k with x_k living on lane k
SRL_G double row_dot(const double row[ND], double x) { double acc = 0.0; row_dot_step<0>(acc, row, x); return acc; }

template <int K> SRL_G void bcast_all_step(double x, double out[ND]) {
    out[K] = bcast<K>(x);
    if constexpr (K + 1 < ND) bcast_all_step<K + 1>(x, out);
}

// out[K] = sum_c a[c] * bcast_K(b[c]) for K = 0..6: Gram rows (a = b = J) and the CRBA products S_K . (Ic S)
template <int K> SRL_G void dot6_bcast_step(const double a[6], const double b[6], double out[ND]) {
    double acc = 0.0;
#pragma unroll
    for (int c = 0; c < 6; c++) fmac_bcast<K>(acc, b[c], a[c]);
    out[K] = acc;
    if constexpr (K + 1 < ND) dot6_bcast_step<K + 1>(a, b, out);
}

template <int K> SRL_G void transpose_upper_step(const Lane &L, const double low[ND], double M[ND]) {
    // M[k] for k > l comes from lane k's low[l]: every lane j < K picks lane K's low[j]
#pragma unroll
    for (int j = 0; j < K; j++) fmac_bcast<K>(M[K], low[j], L.e[j]);
    if constexpr (K + 1 < ND) transpose_upper_step<K + 1>(L, low, M);
}

// ------------------------------------------------------------------ PGS sweeps
struct Rows {
    double acc0;               // the couplings to rows that come LATER in the first sweep, at their initial impulse 0
    double cs;                 // scaled constant term of the own row
    double n[GL];              // scaled couplings -a_rk S_k / (a_rr S_r), n[own] = 0 (the free path touches 0..6 and 8..10 only)
    double diag;               // a_rr
    double lo, S;              // lambda = lo + S u
    double jb;                 // the row's Jacobian entry on the button glider
};

// fast path: no generic row in the wavefront.  The button's three scalar rows (lanes 8, 9, 10) are decoupled from the
// arm rows, so arm row j and button row 8 + j are updated by the same instructions for j < 3.  One sweep = 24 VALU
// instructions; the whole sweep is one asm statement (the compiler pads every asm boundary with wait states).
SRL_G void pgs_sweep_free(double &acc, double cs, const double n[GL], const double e[ND], double e0, double e1, double e2, double ep_first) {
#if SRL_G_DEVICE
    double t;
#define SRL_ROW(J, NJ, EP) "v_add_f64 %1, %2, %0 clamp\n\tv_fma_f64 %0, -%" #EP ", %0, %0\n\ts_nop 0\n\tv_fmac_f64_dpp %0, %1, %" #NJ " row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n\t"
#define SRL_ROWB(J, NJ) "v_fmac_f64_dpp %0, %1, %" #NJ " row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n\t"
    asm volatile(SRL_ROW(0, 3, 13) SRL_ROWB(8, 10)
                 SRL_ROW(1, 4, 14) SRL_ROWB(9, 11)
                 SRL_ROW(2, 5, 15) SRL_ROWB(10, 12)
                 SRL_ROW(3, 6, 16) SRL_ROW(4, 7, 17) SRL_ROW(5, 8, 18) SRL_ROW(6, 9, 19)
                 : "+v"(acc), "=&v"(t)
                 : "v"(cs), "v"(n[0]), "v"(n[1]), "v"(n[2]), "v"(n[3]), "v"(n[4]), "v"(n[5]), "v"(n[6]),      // %2 .. %9
                   "v"(n[kBM]), "v"(n[kBLo]), "v"(n[kBHi]),                                                      // %10 .. %12
                   "v"(ep_first), "v"(e0), "v"(e1), "v"(e2), "v"(e[3]), "v"(e[4]), "v"(e[5]));                   // %13 .. %19
#undef SRL_ROW
#undef SRL_ROWB
#else
    pgs_row2<0, kBM>(acc, cs, n[0], n[kBM], ep_first);
    pgs_row2<1, kBLo>(acc, cs, n[1], n[kBLo], e0);
    pgs_row2<2, kBHi>(acc, cs, n[2], n[kBHi], e1);
    pgs_row<3>(acc, cs, n[3], e2); pgs_row<4>(acc, cs, n[4], e[3]); pgs_row<5>(acc, cs, n[5], e[4]); pgs_row<6>(acc, cs, n[6], e[5]);
#endif
}
SRL_G double pgs_sweeps_free(const Lane &L, const Rows &r) {
    double acc = r.acc0, u = 0.0, t;
    const double e0 = L.e[0] + (L.l == kBM ? 1.0 : 0.0), e1 = L.e[1] + (L.l == kBLo ? 1.0 : 0.0), e2 = L.e[2] + (L.l == kBHi ? 1.0 : 0.0);
    pgs_sweep_free(acc, r.cs, r.n, L.e, e0, e1, e2, 0.0);          // first sweep: nothing to reset yet
#ifdef SRL_GDBG_FIXPOINT      // host harness only: at which sweep does the iteration reach a bitwise fixed point?
    { int fixed_at = -1; double prev = acc;
      for (int it = 1; it < kSolverIters - 1; it++) { pgs_sweep_free(acc, r.cs, r.n, L.e, e0, e1, e2, L.e[6]); const bool same = !gany(acc != prev); if (same && fixed_at < 0) fixed_at = it; if (!same) fixed_at = -1; prev = acc; }
      if (L.l == 0) SRL_GDBG_FIXPOINT(fixed_at); }
#else
    for (int it = 1; it < kSolverIters - 1; it++) pgs_sweep_free(acc, r.cs, r.n, L.e, e0, e1, e2, L.e[6]);
#endif
    // last sweep row by row: every lane keeps the value of its own row
    t = pgs_row2<0, kBM>(acc, r.cs, r.n[0], r.n[kBM], L.e[6]);  u = fma(e0, t, u);
    t = pgs_row2<1, kBLo>(acc, r.cs, r.n[1], r.n[kBLo], e0);    u = fma(e1, t, u);
    t = pgs_row2<2, kBHi>(acc, r.cs, r.n[2], r.n[kBHi], e1);    u = fma(e2, t, u);
    t = pgs_row<3>(acc, r.cs, r.n[3], e2);                      u = fma(L.e[3], t, u);
    t = pgs_row<4>(acc, r.cs, r.n[4], L.e[3]);                  u = fma(L.e[4], t, u);
    t = pgs_row<5>(acc, r.cs, r.n[5], L.e[4]);                  u = fma(L.e[5], t, u);
    t = pgs_row<6>(acc, r.cs, r.n[6], L.e[5]);                  u = fma(L.e[6], t, u);
    return u;
}

// A generic row (slot G, lane kGenLane[G]) in one of the two phases of the general sweep.  act = 1 on lanes whose row
// belongs to this phase: the others broadcast nothing and keep their accumulator.
template <int G, bool LAST> SRL_G void pgs_generic_phase(const Lane &L, const Rows &r, double &acc, double &u, double &ep, uint32_t wave_slots, double act) {
    if (wave_slots & (1u << G)) {            // wave-uniform: some env of this wavefront has a row in slot G
        constexpr int J = kGenLane[G];
        // (1 - act) * 1e300 pushes the residual of a row outside its phase below 0: it broadcasts t = 0
        const double t = pgs_row<J>(acc, r.cs - (1.0 - act) * 1e300, r.n[J], ep);
        ep = (L.l == J ? 1.0 : 0.0) * act;
        if (LAST) u = fma(ep, t, u);
    }
    if constexpr (G + 1 < kMaxGenRows) pgs_generic_phase<G + 1, LAST>(L, r, acc, u, ep, wave_slots, act);
}

// general sweep: arm motors, button motor, [arm joint limits], button stops, [contacts] — Bullet's row order.
template <bool LAST> SRL_G void pgs_sweep_general(const Lane &L, const Rows &r, double &acc, double &u, double &ep, uint32_t wave_slots, bool has_lim,
                                                  double in_lim, double in_con, double ebm, double eblo, double ebhi) {
    double t;
    t = pgs_row<0>(acc, r.cs, r.n[0], ep);       if (LAST) u = fma(L.e[0], t, u);
    t = pgs_row<1>(acc, r.cs, r.n[1], L.e[0]);   if (LAST) u = fma(L.e[1], t, u);
    t = pgs_row<2>(acc, r.cs, r.n[2], L.e[1]);   if (LAST) u = fma(L.e[2], t, u);
    t = pgs_row<3>(acc, r.cs, r.n[3], L.e[2]);   if (LAST) u = fma(L.e[3], t, u);
    t = pgs_row<4>(acc, r.cs, r.n[4], L.e[3]);   if (LAST) u = fma(L.e[4], t, u);
    t = pgs_row<5>(acc, r.cs, r.n[5], L.e[4]);   if (LAST) u = fma(L.e[5], t, u);
    t = pgs_row<6>(acc, r.cs, r.n[6], L.e[5]);   if (LAST) u = fma(L.e[6], t, u);
    t = pgs_row<kBM>(acc, r.cs, r.n[kBM], L.e[6]); if (LAST) u = fma(ebm, t, u);
    ep = ebm;
    if (has_lim) pgs_generic_phase<0, LAST>(L, r, acc, u, ep, wave_slots, in_lim);
    t = pgs_row<kBLo>(acc, r.cs, r.n[kBLo], ep);   if (LAST) u = fma(eblo, t, u);
    t = pgs_row<kBHi>(acc, r.cs, r.n[kBHi], eblo); if (LAST) u = fma(ebhi, t, u);
    ep = ebhi;
    pgs_generic_phase<0, LAST>(L, r, acc, u, ep, wave_slots, in_con);
}
// The common contact case — no joint-limit row in the wavefront and at most two contact rows per env (slots 0, 1 = lanes 7,
// 11): arm motors, button motor and stops, contacts, one asm statement per sweep like the free path.  TWO = slot 1 in use.
template <bool TWO>
SRL_G void pgs_sweep_contact(double &acc, double cs, const double n[GL], const double e[ND], double ebm, double eblo, double ebhi, double eg0,
                             double ep_first) {
#if SRL_G_DEVICE
    double t;
#define SRL_ROW(J, NJ, EP) "v_add_f64 %1, %2, %0 clamp\n\tv_fma_f64 %0, -%" #EP ", %0, %0\n\ts_nop 0\n\tv_fmac_f64_dpp %0, %1, %" #NJ " row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n\t"
    if constexpr (TWO) {
        asm volatile(SRL_ROW(0, 3, 15) SRL_ROW(1, 4, 16) SRL_ROW(2, 5, 17) SRL_ROW(3, 6, 18) SRL_ROW(4, 7, 19) SRL_ROW(5, 8, 20) SRL_ROW(6, 9, 21)
                     SRL_ROW(8, 10, 22) SRL_ROW(9, 11, 23) SRL_ROW(10, 12, 24) SRL_ROW(7, 13, 25) SRL_ROW(11, 14, 26)
                     : "+v"(acc), "=&v"(t)
                     : "v"(cs), "v"(n[0]), "v"(n[1]), "v"(n[2]), "v"(n[3]), "v"(n[4]), "v"(n[5]), "v"(n[6]),      // %2 .. %9
                       "v"(n[kBM]), "v"(n[kBLo]), "v"(n[kBHi]), "v"(n[7]), "v"(n[11]),                              // %10 .. %14
                       "v"(ep_first), "v"(e[0]), "v"(e[1]), "v"(e[2]), "v"(e[3]), "v"(e[4]), "v"(e[5]), "v"(e[6]),  // %15 .. %22
                       "v"(ebm), "v"(eblo), "v"(ebhi), "v"(eg0));                                                    // %23 .. %26
    } else {
        asm volatile(SRL_ROW(0, 3, 14) SRL_ROW(1, 4, 15) SRL_ROW(2, 5, 16) SRL_ROW(3, 6, 17) SRL_ROW(4, 7, 18) SRL_ROW(5, 8, 19) SRL_ROW(6, 9, 20)
                     SRL_ROW(8, 10, 21) SRL_ROW(9, 11, 22) SRL_ROW(10, 12, 23) SRL_ROW(7, 13, 24)
                     : "+v"(acc), "=&v"(t)
                     : "v"(cs), "v"(n[0]), "v"(n[1]), "v"(n[2]), "v"(n[3]), "v"(n[4]), "v"(n[5]), "v"(n[6]),      // %2 .. %9
                       "v"(n[kBM]), "v"(n[kBLo]), "v"(n[kBHi]), "v"(n[7]),                                          // %10 .. %13
                       "v"(ep_first), "v"(e[0]), "v"(e[1]), "v"(e[2]), "v"(e[3]), "v"(e[4]), "v"(e[5]), "v"(e[6]),  // %14 .. %21
                       "v"(ebm), "v"(eblo), "v"(ebhi));                                                              // %22 .. %24
    }
#undef SRL_ROW
#else
    pgs_row<0>(acc, cs, n[0], ep_first); pgs_row<1>(acc, cs, n[1], e[0]); pgs_row<2>(acc, cs, n[2], e[1]); pgs_row<3>(acc, cs, n[3], e[2]);
    pgs_row<4>(acc, cs, n[4], e[3]); pgs_row<5>(acc, cs, n[5], e[4]); pgs_row<6>(acc, cs, n[6], e[5]);
    pgs_row<kBM>(acc, cs, n[kBM], e[6]); pgs_row<kBLo>(acc, cs, n[kBLo], ebm); pgs_row<kBHi>(acc, cs, n[kBHi], eblo);
    pgs_row<7>(acc, cs, n[7], ebhi);
    if (TWO) pgs_row<11>(acc, cs, n[11], eg0);
#endif
}
template <bool TWO>
SRL_G double pgs_sweeps_contact(const Lane &L, const Rows &r) {
    double acc = r.acc0, u = 0.0, t;
    const double ebm = L.l == kBM ? 1.0 : 0.0, eblo = L.l == kBLo ? 1.0 : 0.0, ebhi = L.l == kBHi ? 1.0 : 0.0;
    const double eg0 = L.l == 7 ? 1.0 : 0.0, eg1 = L.l == 11 ? 1.0 : 0.0, elast = TWO ? eg1 : eg0;
    pgs_sweep_contact<TWO>(acc, r.cs, r.n, L.e, ebm, eblo, ebhi, eg0, 0.0);
    for (int it = 1; it < kSolverIters - 1; it++) pgs_sweep_contact<TWO>(acc, r.cs, r.n, L.e, ebm, eblo, ebhi, eg0, elast);
    t = pgs_row<0>(acc, r.cs, r.n[0], elast);     u = fma(L.e[0], t, u);
    t = pgs_row<1>(acc, r.cs, r.n[1], L.e[0]);    u = fma(L.e[1], t, u);
    t = pgs_row<2>(acc, r.cs, r.n[2], L.e[1]);    u = fma(L.e[2], t, u);
    t = pgs_row<3>(acc, r.cs, r.n[3], L.e[2]);    u = fma(L.e[3], t, u);
    t = pgs_row<4>(acc, r.cs, r.n[4], L.e[3]);    u = fma(L.e[4], t, u);
    t = pgs_row<5>(acc, r.cs, r.n[5], L.e[4]);    u = fma(L.e[5], t, u);
    t = pgs_row<6>(acc, r.cs, r.n[6], L.e[5]);    u = fma(L.e[6], t, u);
    t = pgs_row<kBM>(acc, r.cs, r.n[kBM], L.e[6]);   u = fma(ebm, t, u);
    t = pgs_row<kBLo>(acc, r.cs, r.n[kBLo], ebm);    u = fma(eblo, t, u);
    t = pgs_row<kBHi>(acc, r.cs, r.n[kBHi], eblo);   u = fma(ebhi, t, u);
    t = pgs_row<7>(acc, r.cs, r.n[7], ebhi);         u = fma(eg0, t, u);
    if (TWO) { t = pgs_row<11>(acc, r.cs, r.n[11], eg0); u = fma(eg1, t, u); }
    return u;
}

// wave_slots: bit g = some env of the wavefront uses generic slot g; nlim = joint-limit rows of THIS env (they fill
// the first slots); has_lim: some env of the wavefront has a joint-limit row.
SRL_G double pgs_sweeps_general(const Lane &L, const Rows &r, uint32_t wave_slots, int nlim, bool has_lim) {
    if (!has_lim && wave_slots == 1u) return pgs_sweeps_contact<false>(L, r);
    if (!has_lim && wave_slots == 3u) return pgs_sweeps_contact<true>(L, r);
    double acc = r.acc0, u = 0.0, ep = 0.0;
    // a generic row takes part in the limit phase iff its slot index < nlim, else in the contact phase
    int slot = -1;
#pragma unroll
    for (int g = 0; g < kMaxGenRows; g++) if (L.l == kGenLane[g]) slot = g;
    const double in_lim = (slot >= 0 && slot < nlim) ? 1.0 : 0.0, in_con = (slot >= 0 && slot >= nlim) ? 1.0 : 0.0;
    const double ebm = L.l == kBM ? 1.0 : 0.0, eblo = L.l == kBLo ? 1.0 : 0.0, ebhi = L.l == kBHi ? 1.0 : 0.0;
    for (int it = 0; it < kSolverIters - 1; it++) pgs_sweep_general<false>(L, r, acc, u, ep, wave_slots, has_lim, in_lim, in_con, ebm, eblo, ebhi);
    pgs_sweep_general<true>(L, r, acc, u, ep, wave_slots, has_lim, in_lim, in_con, ebm, eblo, ebhi);
    return u;
}

// ------------------------------------------------------------------ one physics step
// Kuka.applyAction (kuka.py:118-187) + p.stepSimulation(), same semantics as physics_step<1>() of kuka_core.hpp.
// `e` holds the env's scalar state replicated on the 16 lanes (its q / qd / sq / cq arrays are not used here), `g` the
// lane's own joint and frame (valid on entry: grefresh()), jt_own the joint-mode target of the own joint.
template <bool CM>
SRL_G void gphysics_step(Env &e, GState &g, const Lane &L, const Cfg &cfg, double *scratch, const double motor[3], bool joint_mode,
                         double jt_own) {
    const double dt = kDt;
    // ---- spatial joint axis about the world origin: S = [z ; p x z]; link-7 frame replicated
    double S[6], Rt[9], pt[3];
#pragma unroll
    for (int k = 0; k < 3; k++) S[k] = g.R[6 + k] * L.am;
    cross3(g.p, S, S + 3);
    tip_frame(g, Rt, pt);
    // ---- IK target accumulate + clip (kuka.py:134-139), one damped-least-squares step (kuka.py:144-156)
    double qdes = jt_own;
    if (!joint_mode) {
        const int b = (cfg.random_target || cfg.two) ? 0 : 1;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            double v = e.ee[k] + motor[k];
            v = v < kEeBox[b][0][k] ? kEeBox[b][0][k] : v;
            v = v > kEeBox[b][1][k] ? kEeBox[b][1][k] : v;
            e.ee[k] = v;
        }
        double ee[3], dS[6], J[6];
        tip_point(Rt, pt, L.eept, ee);
        {
            double d[3];
#pragma unroll
            for (int k = 0; k < 3; k++) d[k] = ee[k] - g.p[k];
            cross3(S, d, J);
#pragma unroll
            for (int k = 0; k < 3; k++) J[3 + k] = S[k];
        }
#pragma unroll
        for (int k = 0; k < 3; k++) dS[k] = e.ee[k] - ee[k];
        {   // orientation error (same construction as ik_step() of kuka_core.hpp), replicated
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
        // (J^T J + damping I) dtheta = J^T dS: row l on lane l
        double A[ND], bb = 0.0;
        dot6_bcast_step<0>(J, J, A);
        const double damping = cfg.two ? kIkDampingDefault : kIkDamping;
#pragma unroll
        for (int k = 0; k < ND; k++) A[k] = fma(damping, L.e[k], A[k]);
#pragma unroll
        for (int c = 0; c < 6; c++) bb = fma(J[c], dS[c], bb);
        gj_step<0, false>(L, A, bb);
        bb *= L.am;
        double all[ND], maxabs = 0.0;
        bcast_all_step<0>(bb, all);
#pragma unroll
        for (int k = 0; k < ND; k++) maxabs = fmax(maxabs, fabs(all[k]));
        qdes = g.q + bb;
        if (wany(maxabs > kIkMaxAngle)) {                    // rare: the step is rescaled to at most 45 degrees per joint
            const double scale = kIkMaxAngle / maxabs;
            if (maxabs > kIkMaxAngle) qdes = g.q + bb * scale;
        }
    }
    // ---- collision detection at the current poses: lane s < 6 owns gripper sphere s.  The exact sphere-cylinder distances
    //      (square roots, divisions) are only evaluated when some sphere of the wavefront can be within the contact
    //      threshold of the button at all: conservative box / disc rejection first (margin 1e-9 over the threshold).
    double cc[3], n_cap[3], n_base[3], d_cap = 1e30, d_base = 1e30;
    tip_point(Rt, pt, L.sph, cc);
    const bool sphere = L.l < kNSphere;
    const double cap_z0 = e.bz + kGliderOriginZ + e.bq;
    {
        const double reach = L.sph[3] + kContactThreshold + 1e-9, dx = cc[0] - e.bx, dy = cc[1] - e.by, rho2 = dx * dx + dy * dy;
        const double rmax = kBaseRadius + reach;           // the base is the wider cylinder
        const double top = fmax(cap_z0 + kCapHeight, e.bz + kBaseHeight), bottom = fmin(cap_z0, e.bz);
        const bool far = cc[2] - top >= reach || bottom - cc[2] >= reach || rho2 >= rmax * rmax;
        if (wany(sphere && !far)) {
            if (sphere) {
                d_cap = sphere_cylinder(cc, L.sph[3], e.bx, e.by, kCapRadius, cap_z0, cap_z0 + kCapHeight, n_cap);
                d_base = sphere_cylinder(cc, L.sph[3], e.bx, e.by, kBaseRadius, e.bz, e.bz + kBaseHeight, n_base);
            }
        }
    }
    const bool c_cap = sphere && d_cap < kContactThreshold, c_base = sphere && d_base < kContactThreshold;
    e.contact_table = gany(sphere && (cc[2] - L.sph[3] - L.table_z < kContactThreshold)) ? 1 : 0;
    // ---- motor target velocity of the own joint
    const double inv_dt = 1.0 / kDt;
    double target = kArmKp * (qdes - g.q) * inv_dt;
    target = target > kArmMaxVel ? kArmMaxVel : target;
    target = target < -kArmMaxVel ? -kArmMaxVel : target;
    // ---- dynamics in world coordinates: velocities and bias accelerations by prefix sums over the chain
    const double qd = g.qd * L.am;
    double W[ND], tau;
    {
        Masks M;
        make_masks(L.l, M);
        double w[3], vo[3], aw[3], av[3];
#pragma unroll
        for (int k = 0; k < 3; k++) { w[k] = masked_sum(S[k] * qd, M.le); vo[k] = masked_sum(S[3 + k] * qd, M.le); }
        {
            double t0[3], t1[3], t2[3];
            cross3(w, S, t0); cross3(w, S + 3, t1); cross3(vo, S, t2);
#pragma unroll
            for (int k = 0; k < 3; k++) {
                aw[k] = masked_sum(t0[k] * qd, M.le);
                av[k] = masked_sum((t1[k] + t2[k]) * qd, M.le, k == 2 ? -kGravityZ : 0.0);
            }
        }
        // rigid-body inertia of the own link about the world origin: Io (xx xy xz yy yz zz), h = m c, m
        double Io[6], h[3];
        {
            const double *R = g.R;
            double cw[3];
#pragma unroll
            for (int k = 0; k < 3; k++) cw[k] = g.p[k] + R[k] * L.com[0] + R[3 + k] * L.com[1] + R[6 + k] * L.com[2];
            const double m = L.mass, ccs = dot3(cw, cw);
            const double ix = L.in[0], iy = L.in[1], iz = L.in[2];
            Io[0] = ix * R[0] * R[0] + iy * R[3] * R[3] + iz * R[6] * R[6] + m * (ccs - cw[0] * cw[0]);
            Io[1] = ix * R[0] * R[1] + iy * R[3] * R[4] + iz * R[6] * R[7] - m * cw[0] * cw[1];
            Io[2] = ix * R[0] * R[2] + iy * R[3] * R[5] + iz * R[6] * R[8] - m * cw[0] * cw[2];
            Io[3] = ix * R[1] * R[1] + iy * R[4] * R[4] + iz * R[7] * R[7] + m * (ccs - cw[1] * cw[1]);
            Io[4] = ix * R[1] * R[2] + iy * R[4] * R[5] + iz * R[7] * R[8] - m * cw[1] * cw[2];
            Io[5] = ix * R[2] * R[2] + iy * R[5] * R[5] + iz * R[8] * R[8] + m * (ccs - cw[2] * cw[2]);
#pragma unroll
            for (int k = 0; k < 3; k++) h[k] = m * cw[k];
        }
        // link force f = I a + v x* (I v); then suffix sums: total force on the sub-chain, composite inertia
        double Fn[3], Ff[3], Ioc[6], hc[3];
        {
            double n[3], f[3], t0[3], t1[3], fn[3], ff[3];
            sym_mul(Io, aw, n); cross3(h, av, t0); cross3(h, aw, t1);
#pragma unroll
            for (int k = 0; k < 3; k++) { fn[k] = n[k] + t0[k]; ff[k] = L.mass * av[k] - t1[k]; }
            sym_mul(Io, w, n); cross3(h, vo, t0); cross3(h, w, t1);
#pragma unroll
            for (int k = 0; k < 3; k++) { n[k] += t0[k]; f[k] = L.mass * vo[k] - t1[k]; }
            cross3(w, n, t0); cross3(vo, f, t1);
#pragma unroll
            for (int k = 0; k < 3; k++) fn[k] += t0[k] + t1[k];
            cross3(w, f, t0);
#pragma unroll
            for (int k = 0; k < 3; k++) ff[k] += t0[k];
#pragma unroll
            for (int k = 0; k < 3; k++) { Fn[k] = masked_sum(fn[k], M.ge); Ff[k] = masked_sum(ff[k], M.ge); hc[k] = masked_sum(h[k], M.ge); }
#pragma unroll
            for (int k = 0; k < 6; k++) Ioc[k] = masked_sum(Io[k], M.ge);
        }
        tau = -L.damping * qd - (dot3(S, Fn) + dot3(S + 3, Ff));
        // ---- CRBA: M_kl = S_k . (Ic_l S_l) for k <= l on lane l, the upper part by transposition; W = M^-1 in place
        double Fc[6], t0[3], t1[3], low[ND];
        sym_mul(Ioc, S, Fc); cross3(hc, S + 3, t0); cross3(hc, S, t1);
#pragma unroll
        for (int k = 0; k < 3; k++) { Fc[k] += t0[k]; Fc[3 + k] = L.mcomp * S[3 + k] - t1[k]; }
        dot6_bcast_step<0>(Fc, S, low);
#pragma unroll
        for (int k = 0; k < ND; k++) W[k] = low[k] * M.le[k] * L.am;
        transpose_upper_step<1>(L, low, W);
        double unused = 0.0;
        gj_step<0, true>(L, W, unused);
    }
    const double qdd = row_dot(W, tau);
#pragma unroll
    for (int k = 0; k < ND; k++) SRL_GDBG(0, L.l * ND + k, W[k]);
    SRL_GDBG(1, L.l, qdd); SRL_GDBG(2, L.l, tau); SRL_GDBG(3, L.l, qdes); SRL_GDBG(4, L.l, target);
    double qd_new = qd + dt * qdd;            // unconstrained velocity; the solver corrects it below
    e.bqd += dt * kGravityZ;
    // ---- constraint rows (impulse space, A = J W J^T one row per lane)
    const double arm_bound = kArmMaxForce * dt, wb = 1.0 / kCapMass, blim = kLimitMaxImpulse;
    const double bound_bm = e.motor_on ? kButtonMaxForce * dt : kDefaultMotorImpulse;
    const bool is_bm = L.l == kBM, is_blo = L.l == kBLo, is_bhi = L.l == kBHi, is_button = is_bm || is_blo || is_bhi;
    Rows r;
    double rhs = 0.0;                         // desired velocity change of the own row (velocity units)
    double off = 0.0;                         // sum_{k != own} a_rk lo_k
#pragma unroll
    for (int k = 0; k < GL; k++) r.n[k] = 0.0;
    r.lo = 0.0; r.S = 0.0; r.jb = 0.0; r.diag = 0.0;
    if (L.arm) {
        double sumw = 0.0;
#pragma unroll
        for (int k = 0; k < ND; k++) { r.diag = fma(L.e[k], W[k], r.diag); sumw += W[k] * (1.0 - L.e[k]); }
        rhs = target - qd_new; r.lo = -arm_bound; r.S = 2.0 * arm_bound; off = -arm_bound * sumw;
    } else if (is_bm) {
        rhs = (e.motor_on ? kButtonKp * (kButtonTarget - e.bq) * inv_dt : 0.0) - e.bqd;
        r.lo = -bound_bm; r.S = 2.0 * bound_bm; r.jb = 1.0; r.diag = wb;
    } else if (is_blo) {
        const double pen = e.bq - kGliderLower;
        rhs = ((pen > 0 ? -pen * inv_dt : 0.0) - e.bqd) + (pen > 0 ? 0.0 : -pen * kErp * inv_dt);
        r.S = blim; r.jb = 1.0; r.diag = wb; off = wb * -bound_bm;
    } else if (is_bhi) {
        const double pen = kGliderUpper - e.bq;
        rhs = ((pen > 0 ? -pen * inv_dt : 0.0) + e.bqd) + (pen > 0 ? 0.0 : -pen * kErp * inv_dt);
        r.S = blim; r.jb = -1.0; r.diag = wb; off = -wb * -bound_bm;
    }
    // generic rows: joint-limit candidates of the own joint, contact candidates of the own sphere
    const double pen_lo = g.q - L.jlo, pen_hi = L.jhi - g.q;
    const bool lim_lo = L.arm && pen_lo <= kLimitActivationVel * dt, lim_hi = L.arm && pen_hi <= kLimitActivationVel * dt;
    e.contact_button = gany(c_cap) ? 1 : 0;
    uint32_t wave_slots = 0; int nlim = 0; bool has_lim = false;
    double agen[kMaxGenRows] = {0, 0, 0, 0, 0, 0};       // unscaled couplings of the own row to the generic rows
    const bool any_generic = wany(lim_lo || lim_hi || c_cap || c_base);
#ifdef SRL_GDBG_COUNTS       // host harness only: how often a step carries generic rows
    { const bool gl = gany(lim_lo || lim_hi), gc = gany(c_cap), gb = gany(c_base); if (L.l == 0) SRL_GDBG_COUNTS(any_generic, gl, gc, gb); }
#endif
    if (any_generic) {
        // slot of a candidate = number of candidates before it in Bullet's creation order: limits (joint 0 lower, joint 0
        // upper, joint 1 lower, ...), then contacts (sphere 0 cap, sphere 0 base, sphere 1 cap, ...); the first six are kept
        const uint32_t b_lo = ballot(lim_lo), b_hi = ballot(lim_hi), b_cap = ballot(c_cap), b_base = ballot(c_base);
        const uint32_t below = (1u << L.l) - 1u;
        nlim = __builtin_popcount(b_lo) + __builtin_popcount(b_hi);
        const int ncon = __builtin_popcount(b_cap) + __builtin_popcount(b_base);
        const int s_lo = __builtin_popcount(b_lo & below) + __builtin_popcount(b_hi & below), s_hi = s_lo + (lim_lo ? 1 : 0);
        const int s_cap = nlim + __builtin_popcount(b_cap & below) + __builtin_popcount(b_base & below), s_base = s_cap + (c_cap ? 1 : 0);
        if (nlim > kMaxGenRows) nlim = kMaxGenRows;
        int ngen = nlim + ncon; if (ngen > kMaxGenRows) ngen = kMaxGenRows;
        // every joint axis on every lane (contact Jacobians): only here, off the common path
        double Sz[ND][3], S2[ND][3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            double tz[ND], t2[ND];
            bcast_all_step<0>(S[k], tz); bcast_all_step<0>(S[3 + k], t2);
#pragma unroll
            for (int j = 0; j < ND; j++) { Sz[j][k] = tz[j]; S2[j][k] = t2[j]; }
        }
        // row definitions -> scratch [slot][12]: J[7] Jb desired pos_err lo hi
        double *def = scratch, *wj = scratch + kMaxGenRows * kRowDef;
#define SRL_DEF_ROW(slot_, fillJ, Jb_, des_, perr_, hi_)                                     \
        if ((slot_) < kMaxGenRows) {                                                         \
            double *o = def + (slot_) * kRowDef;                                             \
            fillJ                                                                            \
            o[7] = (Jb_); o[8] = (des_); o[9] = (perr_); o[10] = 0.0; o[11] = (hi_);         \
        }
        if (lim_lo) { SRL_DEF_ROW(s_lo, _Pragma("unroll") for (int j = 0; j < ND; j++) o[j] = L.e[j];, 0.0, pen_lo > 0 ? -pen_lo / dt : 0.0, pen_lo > 0 ? 0.0 : -pen_lo * kErp / dt, blim) }
        if (lim_hi) { SRL_DEF_ROW(s_hi, _Pragma("unroll") for (int j = 0; j < ND; j++) o[j] = -L.e[j];, 0.0, pen_hi > 0 ? -pen_hi / dt : 0.0, pen_hi > 0 ? 0.0 : -pen_hi * kErp / dt, blim) }
#define SRL_CONTACT_J(nrm)                                                                   \
        {                                                                                    \
            double pt3[3];                                                                   \
            _Pragma("unroll") for (int k = 0; k < 3; k++) pt3[k] = cc[k] - L.sph[3] * nrm[k]; \
            _Pragma("unroll") for (int j = 0; j < ND; j++) {                                 \
                double c3[3];                                                                \
                cross3(Sz[j], pt3, c3);                                                      \
                o[j] = dot3(nrm, c3) + dot3(nrm, S2[j]);      /* n . (z_j x (pt - p_j)) */   \
            }                                                                                \
        }
        if (c_cap) { SRL_DEF_ROW(s_cap, SRL_CONTACT_J(n_cap), -n_cap[2], d_cap > 0 ? -d_cap / dt : 0.0, d_cap > 0 ? 0.0 : -d_cap * kErp / dt, 1e10) }
        if (c_base) { SRL_DEF_ROW(s_base, SRL_CONTACT_J(n_base), 0.0, d_base > 0 ? -d_base / dt : 0.0, d_base > 0 ? 0.0 : -d_base * kErp / dt, 1e10) }
#undef SRL_CONTACT_J
#undef SRL_DEF_ROW
        sync_scratch();
        has_lim = wany(nlim > 0);
        int myslot = -1;
#pragma unroll
        for (int gi = 0; gi < kMaxGenRows; gi++) if (L.l == kGenLane[gi]) myslot = gi;
        // W J of every row: lane k < 7 computes (W J_g)_k = a_{k,g}; gathered per row through the scratch
        double Jb[kMaxGenRows];
#pragma unroll
        for (int gi = 0; gi < kMaxGenRows; gi++) {
            const bool have = gi < ngen;
            Jb[gi] = have ? def[gi * kRowDef + 7] : 0.0;
            double wjk = 0.0;
#pragma unroll
            for (int j = 0; j < ND; j++) wjk = fma(W[j], have ? def[gi * kRowDef + j] : 0.0, wjk);
            if (L.arm) { agen[gi] = wjk; wj[gi * ND + L.l] = wjk; }
            else if (is_button) agen[gi] = r.jb * wb * Jb[gi];
            wave_slots |= wany(have) ? (1u << gi) : 0u;
        }
        sync_scratch();
        double aarm[ND] = {0, 0, 0, 0, 0, 0, 0};       // a generic row's couplings to the arm rows
        const bool mine = myslot >= 0 && myslot < ngen;
        if (mine) {
            const double *o = def + myslot * kRowDef, *mywj = wj + myslot * ND;
            r.jb = o[7];
#pragma unroll
            for (int k = 0; k < ND; k++) aarm[k] = mywj[k];
#pragma unroll
            for (int gi = 0; gi < kMaxGenRows; gi++) {
                double c = r.jb * wb * Jb[gi];
#pragma unroll
                for (int j = 0; j < ND; j++) c = fma(o[j], gi < ngen ? wj[gi * ND + j] : 0.0, c);
                agen[gi] = gi < ngen ? c : 0.0;
                if (gi == myslot) r.diag = c;
            }
            r.lo = o[10]; r.S = o[11] - o[10];
            rhs = o[8] + o[9] - o[7] * e.bqd;    // the arm part of the row velocity is subtracted below (needs every lane's qd_new)
            // couplings to rows with a non-zero lower bound: the arm motors (-arm_bound) and the button motor (-bound_bm)
            off = r.jb * wb * -bound_bm;
#pragma unroll
            for (int k = 0; k < ND; k++) off = fma(aarm[k], -arm_bound, off);
        }
        // J . qd_new of every generic row (qd_new lives one joint per lane): replicated dot products
        {
            double qall[ND];
            bcast_all_step<0>(qd_new, qall);
            if (mine) {
                double sdot = 0.0;
#pragma unroll
                for (int j = 0; j < ND; j++) sdot = fma(def[myslot * kRowDef + j], qall[j], sdot);
                rhs -= sdot;
            }
        }
        sync_scratch();                          // scratch is reused by the next step
        // scaled couplings: n_rk = -a_rk S_k / (a_rr S_r)
        const double S0 = bcast<kGenLane[0]>(r.S), S1 = bcast<kGenLane[1]>(r.S), S2g = bcast<kGenLane[2]>(r.S), S3 = bcast<kGenLane[3]>(r.S),
                     S4 = bcast<kGenLane[4]>(r.S), S5 = bcast<kGenLane[5]>(r.S);
        const bool live = r.S > 0.0 && r.diag > 0.0;
        const double inv = live ? 1.0 / (r.diag * r.S) : 0.0;
        const double Sg[kMaxGenRows] = {S0, S1, S2g, S3, S4, S5};
#pragma unroll
        for (int gi = 0; gi < kMaxGenRows; gi++) r.n[kGenLane[gi]] = (mine && gi == myslot) ? 0.0 : -agen[gi] * Sg[gi] * inv;
        if (mine) {
#pragma unroll
            for (int k = 0; k < ND; k++) r.n[k] = -aarm[k] * (2.0 * arm_bound) * inv;
            r.n[kBM] = -(r.jb * wb) * (2.0 * bound_bm) * inv; r.n[kBLo] = -(r.jb * wb) * blim * inv; r.n[kBHi] = (r.jb * wb) * blim * inv;
        }
    }
    // ---- scale the arm / button rows to u in [0, 1]:  x_r = cs_r + sum_k n_rk u_k
    {
        const bool live = r.S > 0.0 && r.diag > 0.0;
        const double inv = live ? rcp(r.diag * r.S) : 0.0;
        // -lo / S: 1/2 for the symmetric rows (arm motors, button motor), 0 for the unilateral ones
        r.cs = live ? (rhs - off) * inv + ((L.arm || is_bm) ? 0.5 : 0.0) : 0.0;
        if (L.arm) {
#pragma unroll
            for (int k = 0; k < ND; k++) r.n[k] = -(W[k] * (1.0 - L.e[k])) * (2.0 * arm_bound) * inv;
        } else if (is_button) {
            r.n[kBM] = is_bm ? 0.0 : -(r.jb * wb) * (2.0 * bound_bm) * inv;
            r.n[kBLo] = is_blo ? 0.0 : -(r.jb * wb) * blim * inv;
            r.n[kBHi] = is_bhi ? 0.0 : (r.jb * wb) * blim * inv;
        }
        // lambda starts at 0, i.e. u_k = -lo_k / S_k = 1/2 for the symmetric rows (arm motors; the button motor couples to
        // no row that comes before it): what arm row l sees of the arm rows behind it during the first sweep
        r.acc0 = 0.0;
        if (L.arm) {
#pragma unroll
            for (int k = 0; k < ND; k++) r.acc0 = fma(r.n[k] * (k > L.l ? 1.0 : 0.0), 0.5, r.acc0);
        }
    }
    const double u = any_generic ? pgs_sweeps_general(L, r, wave_slots, nlim, has_lim) : pgs_sweeps_free(L, r);
    const double lam = r.lo + r.S * u;
    SRL_GDBG(5, L.l, lam);
#ifdef SRL_GDBG_CLAMPS        // host harness only: how often does an arm motor row end the solve at its bound?
    { const bool cl = gany(L.arm && (u <= 0.0 || u >= 1.0));
      // a-priori test (energy norm of the Gauss-Seidel error is non-increasing): can a clamp trigger at all?
      double lamstar = 0.0; { double Mrow_dummy = 0.0; (void)Mrow_dummy; }
      if (L.l == 0) SRL_GDBG_CLAMPS(cl); }
#endif
    // ---- velocity change: arm lane i gets sum_r a_ir lambda_r = diag_i (lambda_i - S_i sum_{k != i} n_ik lambda_k / S_k),
    //      the glider sum_r jb_r lambda_r / m
    double dv = 0.0, dvb = 0.0;
    {
        const double v = r.S > 0.0 ? lam / r.S : 0.0, pb = r.jb * lam * wb;
        double acc = 0.0;
#define SRL_ACC(K) fmac_bcast<K>(acc, v, r.n[K]);
        SRL_ACC(0) SRL_ACC(1) SRL_ACC(2) SRL_ACC(3) SRL_ACC(4) SRL_ACC(5) SRL_ACC(6)
        dvb = bcast<kBM>(pb) + bcast<kBLo>(pb) + bcast<kBHi>(pb);
        if (any_generic) {
            SRL_ACC(7) SRL_ACC(11) SRL_ACC(12) SRL_ACC(13) SRL_ACC(14) SRL_ACC(15)
            dvb += bcast<7>(pb) + bcast<11>(pb) + bcast<12>(pb) + bcast<13>(pb) + bcast<14>(pb) + bcast<15>(pb);
        }
#undef SRL_ACC
        dv = r.diag * (lam - r.S * acc);
    }
    // ---- semi-implicit Euler, refresh sin/cos, frames and the gripper position
    if (L.arm) { g.qd = qd_new + dv; g.q += dt * g.qd; }
    e.bqd += dvb;
    e.bq += dt * e.bqd;
    grefresh<CM>(L, g, e);
}

// ------------------------------------------------------------------ env level (mirrors kuka_env.hpp for a lane group)
// RNG adaptor for generators whose state lives in HBM (MT19937): lane 0 draws, the row receives the value.
template <class R> struct Lane0Rng {
    R *r; bool own;        // own: this lane is lane 0 of a valid env
    SRL_G double double01() { double v = 0.0; if (own) v = r->double01(); return bcast<0>(v); }
    SRL_G double uniform(double a, double b) { double v = 0.0; if (own) v = r->uniform(a, b); return bcast<0>(v); }
    SRL_G double normal(double a, double b) { double v = 0.0; if (own) v = r->normal(a, b); return bcast<0>(v); }
    SRL_G uint32_t bounded(uint32_t m) { double v = 0.0; if (own) v = (double)r->bounded(m); return (uint32_t)bcast<0>(v); }
};

// numpy RandomState (MT19937, state words in HBM: rng.hpp Mt19937View) for a lane group.  The index and the cached Gaussian are
// replicated on the 16 lanes and every draw is computed by all of them (no broadcast of results).  Raw state words are fetched
// SIXTEEN AT A TIME — one load instruction, lane l takes word idx + l — and handed out by a row shuffle: a per-word load inside the
// step loop would wait for the previous steps' output stores on every draw (gfx9 counts loads and stores in one in-order
// counter).  The twist regenerates the 624 words in 39 blocks of 16, word b + l on lane l (3 loads + 1 store per lane and block
// instead of lane 0's 33 + 16): same blocks, same read-before-write order as Mt19937::twist, so the stream is the sequential
// generator's bit for bit.  (Memory operations of one wavefront are performed in order: the wavefront-scope fences of
// sync_scratch() order the blocks; on the host harness they are the fibers' lockstep points.)
struct GroupMt {
    Mt19937 m;
    uint32_t win;            // raw (untempered) word m.idx - pos + lane
    int pos, cnt;            // words of the window handed out / fetched
    SRL_G void load(const Mt19937View &v, int64_t env) { m.load(v, env); win = 0u; pos = 0; cnt = 0; }
    SRL_G void store(const Mt19937View &v, int64_t env) const { m.store(v, env); }      // (one lane of a valid env)
    SRL_G void twist() {
        const int l = lane_id();
        for (int b = 0; b < MT_N; b += GL) {                      // 624 = 39 x 16
            const int k = b + l, k1 = k + 1 >= MT_N ? k + 1 - MT_N : k + 1, km = k + MT_M >= MT_N ? k + MT_M - MT_N : k + MT_M;
            const uint32_t a0 = m.at(k), a1 = m.at(k1), mm = m.at(km);
            sync_scratch();
            const uint32_t y = (a0 & 0x80000000u) | (a1 & 0x7fffffffu);
            uint32_t v = mm ^ (y >> 1);
            if (y & 1u) v ^= 0x9908b0dfu;
            m.at(k) = v;
            sync_scratch();
        }
        m.idx = 0;
    }
    SRL_G uint32_t u32() {
        if (pos >= cnt) {
            if (m.idx >= MT_N) twist();
            const int l = lane_id();
            cnt = MT_N - m.idx < GL ? MT_N - m.idx : GL;
            win = l < cnt ? m.at(m.idx + l) : 0u;
            pos = 0;
        }
        const uint32_t y = (uint32_t)shfl((double)win, pos);      // (a uint32 is exact in a double)
        pos++; m.idx++;
        return mt_temper(y);
    }
    SRL_G double double01() { return mt_double01(*this); }
    SRL_G double normal(double loc, double scale) { return mt_normal(*this, m.has_g, m.g, loc, scale); }
    SRL_G double uniform(double low, double high) { return mt_uniform(*this, low, high); }
    SRL_G uint32_t bounded(uint32_t rng) { return mt_bounded(*this, rng); }
};

// Counter-based env stream (Philox) for a lane group.  Every lane holds the same key / counter; the expensive draw — the
// per-step Gaussian noise (log, sqrt, cos in float64) — is produced 16 counters at a time, one per lane, and handed out by a
// row shuffle, so its cost is shared by 16 steps instead of being replayed on all 16 lanes every step.  Values are the
// sequential stream's bit for bit (a deviate is a pure function of key and counter).
struct GroupPhilox {
    Philox p;
    uint64_t base;       // counter the batch starts at
    double z;            // std_normal() at counter base + lane
    bool have;
    SRL_G void init(uint32_t k0, uint32_t k1, uint64_t ctr) { p.k0 = k0; p.k1 = k1; p.ctr = ctr; p.stream = 0; base = 0; z = 0.0; have = false; }
    SRL_G double double01() { return p.double01(); }
    SRL_G double uniform(double a, double b) { return p.uniform(a, b); }
    SRL_G uint32_t bounded(uint32_t m) { return p.bounded(m); }
    SRL_G double normal(double loc, double scale) {
#pragma clang fp contract(off)
        if (wany(!have || p.ctr - base >= (uint64_t)GL)) {       // every row of the wavefront refills together
            base = p.ctr;
            Philox q = p; q.ctr = base + (uint64_t)lane_id();
            z = q.std_normal();
            have = true;
        }
        const double zz = shfl(z, (int)(p.ctr - base));
        p.ctr++;
        return loc + scale * zz;
    }
};
// the synthetic agent's discrete action stream, batched the same way
struct GroupActions {
    Philox p;            // stream 1
    uint64_t base;
    double a;            // bounded(5) at counter base + lane (as a double: it travels by the same shuffle)
    bool have;
    SRL_G void init(uint32_t k0, uint32_t k1, uint64_t ctr) { p.k0 = k0; p.k1 = k1; p.ctr = ctr; p.stream = 1; base = 0; a = 0.0; have = false; }
    SRL_G int next(uint32_t m) {
        if (wany(!have || p.ctr - base >= (uint64_t)GL)) {
            base = p.ctr;
            Philox q = p; q.ctr = base + (uint64_t)lane_id();
            a = (double)q.bounded(m);
            have = true;
        }
        const int v = (int)shfl(a, (int)(p.ctr - base));
        p.ctr++;
        return v;
    }
};

// arm / glider part of a packed start state (pack_start() layout) -> lane group
SRL_G void gunpack_start(Env &e, GState &g, const Lane &L, const double *o) {
    if (L.arm) { g.q = o[L.l]; g.qd = o[7 + L.l]; g.sq = o[14 + L.l]; g.cq = o[21 + L.l]; }
    else { g.q = 0.0; g.qd = 0.0; g.sq = 0.0; g.cq = 1.0; }
#pragma unroll
    for (int k = 0; k < 3; k++) { e.ee[k] = o[28 + k]; e.grip[k] = o[33 + k]; }
    e.bq = o[31]; e.bqd = o[32];
}

// KukaButtonGymEnv.reset for one lane group (same draws, same table as reset_env<1>).  JOINTS = the continuous joint-space
// action mode, whose five init actions are integrated here: a compile-time switch so that the other modes carry a single
// copy of the physics step.
template <bool JOINTS, bool CM, class R>
SRL_G void genv_reset(Env &e, GState &g, const Lane &L, const Cfg &cfg, double *scratch, R &rng, const double *starts, const double *settled,
                      double *objs, int64_t objs_stride) {
#pragma clang fp contract(off)
    ResetDraw d;
    reset_draw<1>(cfg, L.l == 0 ? objs : nullptr, objs_stride, rng, d);
    e.motor_on = 0; e.contact_button = 0; e.contact_table = 0;
    // baked model: the start-state table; runtime model table: the init actions are integrated from the settled state
    gunpack_start(e, g, L, (JOINTS || CM) ? settled : starts + (int64_t)d.idx * kStartDoubles);
    e.bx = d.bx; e.by = d.by; e.bz = L.base_z;
    gfk<CM>(L, g);
    if constexpr (JOINTS) {
        const double motor[3] = {0, 0, 0};
        for (int k = 0; k < kNInitActions; k++) {
            const double jt = L.q0 + kDeltaTheta * d.g[k];
            gphysics_step<CM>(e, g, L, cfg, scratch, motor, true, jt);
        }
    } else if constexpr (CM) {
        const int base = cfg.is_discrete ? 6 : 2;
        int rem = d.idx;
        for (int k = 0; k < kNInitActions; k++) {
            double motor[3];
            init_action_motor(cfg, rem % base, motor);
            gphysics_step<CM>(e, g, L, cfg, scratch, motor, false, L.q0);
            rem /= base;
        }
    }
    reset_finish<1>(e, d, L.base_z);
}

// KukaButtonGymEnv.step + step2 for one lane group.  ca3: the Cartesian action (replicated), ca_own: the own joint's action.
template <bool CM, class R>
SRL_G double genv_step(Env &e, GState &g, const Lane &L, const Cfg &cfg, double *scratch, R &rng, int action, const float *ca3, float ca_own,
                       bool *done) {
    StepCmd c;
    step_command(e, cfg, rng, action, ca3, c);
    const double jt = joint_target(c, ca_own, L.q0);
    for (int rep = 0; rep < cfg.action_repeat; rep++) {
        gphysics_step<CM>(e, g, L, cfg, scratch, c.motor, c.joint_mode, jt);
        if (termination(e, cfg)) break;
        e.counter += 1;
    }
    const double reward = reward_fn(e, cfg);
    *done = termination(e, cfg);
    return reward;
}

}  // namespace grp
}  // namespace kuka
}  // namespace srl
