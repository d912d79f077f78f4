// kuka_tree_kernels.hpp — the rollout / reset kernel templates of the full-model lane-group stepper, shared by the translation units that
// instantiate them: kuka_tree.hip (every Kuka env) and kuka_tree_rb.hip (KukaRandButtonGymEnv: RB = 1, the free bodies of
// kuka_tree.hpp's free-body section ride on lanes 0..10).  Launch geometry of the lane-group family: 16 lanes = one DPP row per env,
// one wavefront (4 envs) per workgroup — a 4096-env batch is 1024 wavefronts, one per SIMD of the MI355X.
#pragma once
#include "kuka_device.hpp"
#include "kuka_tree.hpp"

namespace srl {
using namespace kuka;

namespace {
using tree::TLane;
using tree::NJ;
constexpr int kTS = tree::kTreeScratchDoubles, kTStart = tree::kTreeStartDoubles;

// the wavefront's lane-constant table in LDS (kuka_tree.hpp: lane_store / lane_view).  It is derived from the model table ONCE per
// model (kuka_tree_table_k -> KukaState::ttable in HBM: the ancestor / descendant walks cost ~0.1 ms, far too much for the
// single-step launches of the per-step API); a launch only copies its 5.6 KB.
struct LaneId { int l; bool jnt; double q0, base_z; };
__device__ __forceinline__ void build_lane_table(LaneId &L, const double *ttable, double *tab) {
    // every load of the copy in flight before the first wait (a rolled copy loop waits out one global-memory round trip per 512
    // bytes: eleven of them, ~7 us of a 38-us single-step launch)
    constexpr int kCopyIters = (tree::kLaneTableDoubles + kGroupBlock - 1) / kGroupBlock;
    double row[kCopyIters];
#pragma unroll
    for (int k = 0; k < kCopyIters; k++) { const int i = (int)threadIdx.x + k * kGroupBlock; row[k] = i < tree::kLaneTableDoubles ? ttable[i] : 0.0; }
#pragma unroll
    for (int k = 0; k < kCopyIters; k++) { const int i = (int)threadIdx.x + k * kGroupBlock; if (i < tree::kLaneTableDoubles) tab[i] = row[k]; }
    __syncthreads();
    L.l = (int)(threadIdx.x & (grp::GL - 1)); L.jnt = L.l < NJ;
    L.q0 = tab[tree::LT_Q0 * grp::GL + L.l]; L.base_z = tab[tree::LT_COUNT * grp::GL + tree::LS_BASEZ];
}
// env scalars replicated on the row, the own joint per joint lane
__device__ __forceinline__ void tload(const KukaState &s, int64_t n, int e, const LaneId &L, Env &v, grp::GState &g, bool two) {
#pragma unroll
    for (int k = 0; k < 3; k++) { v.ee[k] = s.d[(D_EE + k) * n + e]; v.bpos[k] = s.d[(D_BPOS + k) * n + e]; v.grip[k] = s.d[(D_GRIP + k) * n + e]; }
    v.bq = s.d[D_BQ * n + e]; v.bqd = s.d[D_BQD * n + e]; v.bx = s.d[D_BX * n + e]; v.by = s.d[D_BY * n + e];
    v.bz = s.d[D_BZ * n + e]; v.bspeed = s.d[D_BSPEED * n + e];
    v.motor_on = s.i[I_MOTOR * n + e]; v.contact_button = s.i[I_CB * n + e]; v.contact_table = s.i[I_CT * n + e];
    v.counter = s.i[I_COUNTER * n + e]; v.n_contacts = s.i[I_NCONTACT * n + e]; v.n_outside = s.i[I_NOUT * n + e];
    v.terminated = s.i[I_TERM * n + e]; v.ikx = s.i[I_IKX * n + e];
    if (two) {                               // Kuka2ButtonGymEnv: second glider, goal switching
        v.b2q = s.d[D_B2Q * n + e]; v.b2qd = s.d[D_B2QD * n + e]; v.b2x = s.d[D_B2X * n + e]; v.b2y = s.d[D_B2Y * n + e];
        v.goal_id = s.i[I_GOAL * n + e]; v.n_contacts2 = s.i[I_NCONTACT2 * n + e]; v.contact_body1 = 0; v.contact_body2 = 0;
    }
    const int j = L.jnt ? L.l : 0;
    g.q = s.d[tree_plane(D_Q, D_GQ, j) * n + e]; g.qd = s.d[tree_plane(D_QD, D_GQD, j) * n + e];
    g.sq = s.d[tree_plane(D_SQ, D_GSQ, j) * n + e]; g.cq = s.d[tree_plane(D_CQ, D_GCQ, j) * n + e];
    if (!L.jnt) { g.q = 0.0; g.qd = 0.0; g.sq = 0.0; g.cq = 1.0; }
}
__device__ __forceinline__ void tstore(const KukaState &s, int64_t n, int e, const LaneId &L, const Env &v, const grp::GState &g, bool valid, bool two) {
    if (valid && L.jnt) {
        s.d[tree_plane(D_Q, D_GQ, L.l) * n + e] = g.q; s.d[tree_plane(D_QD, D_GQD, L.l) * n + e] = g.qd;
        s.d[tree_plane(D_SQ, D_GSQ, L.l) * n + e] = g.sq; s.d[tree_plane(D_CQ, D_GCQ, L.l) * n + e] = g.cq;
    }
    if (valid && L.l == 0) {
#pragma unroll
        for (int k = 0; k < 3; k++) { s.d[(D_EE + k) * n + e] = v.ee[k]; s.d[(D_BPOS + k) * n + e] = v.bpos[k]; s.d[(D_GRIP + k) * n + e] = v.grip[k]; }
        s.d[D_BQ * n + e] = v.bq; s.d[D_BQD * n + e] = v.bqd; s.d[D_BX * n + e] = v.bx; s.d[D_BY * n + e] = v.by;
        s.d[D_BZ * n + e] = v.bz; s.d[D_BSPEED * n + e] = v.bspeed;
        s.i[I_MOTOR * n + e] = v.motor_on; s.i[I_CB * n + e] = v.contact_button; s.i[I_CT * n + e] = v.contact_table;
        s.i[I_COUNTER * n + e] = v.counter; s.i[I_NCONTACT * n + e] = v.n_contacts; s.i[I_NOUT * n + e] = v.n_outside;
        s.i[I_TERM * n + e] = v.terminated; s.i[I_IKX * n + e] = v.ikx;
        if (two) {
            s.d[D_B2Q * n + e] = v.b2q; s.d[D_B2QD * n + e] = v.b2qd; s.d[D_B2X * n + e] = v.b2x; s.d[D_B2Y * n + e] = v.b2y;
            s.i[I_GOAL * n + e] = v.goal_id; s.i[I_NCONTACT2 * n + e] = v.n_contacts2;
        }
    }
}

// KukaRandButton free bodies: lane k < 11 owns body k.  Dynamic state in the rb planes [(6 k + c)][n] (x y z vx vy vz); the reset draws
// (x, y, kept) of the ten distractors in the objs planes (srlhip_get_state KUKA_OBJECTS), from which the type hash and the kick
// direction derive.
__device__ __forceinline__ void tload_body(const KukaState &s, int64_t n, int e, int l, tree::RBody &B) {
    const int k = l < tree::kRbN ? l : 0;
#pragma unroll
    for (int c = 0; c < 3; c++) { B.x[c] = s.rb[(6 * k + c) * n + e]; B.v[c] = s.rb[(6 * k + 3 + c) * n + e]; }
    const int ko = l < 10 ? l : 0;
    B.ox = s.objs[(3 * ko) * n + e]; B.oy = s.objs[(3 * ko + 1) * n + e];
    B.on = l < 10 ? s.objs[(3 * ko + 2) * n + e] != 0.0 : l == 10;
    B.type = l < 10 ? tree::rb_type_of(B.ox, B.oy) : l == 10 ? 3 : 2;
    if (l >= 10) { B.ox = 0.0; B.oy = 0.0; }
}
__device__ __forceinline__ void tstore_body(const KukaState &s, int64_t n, int e, int l, const tree::RBody &B, bool valid) {
    if (valid && l < tree::kRbN) {
#pragma unroll
        for (int c = 0; c < 3; c++) { s.rb[(6 * l + c) * n + e] = B.x[c]; s.rb[(6 * l + 3 + c) * n + e] = B.v[c]; }
    }
}

// T consecutive VecEnv steps per launch.  GIVEN: the caller supplies the actions (a compile-time switch: a possible action load
// inside the step loop makes every step wait for the previous step's output stores — gfx9 counts loads and stores together).
// SPEC = 1: the reference's DEFAULT KukaButtonGymEnv configuration (discrete actions, static button, ground-truth observation, force_down,
// action_repeat 1, sparse reward, auto-reset: kuka_button_gym_env.py:93-98 ctor defaults) AND the model table's default solver details
// (solver_detail = 0) as compile-time constants — the run-time
// configuration tests of the env logic and of the step's branches fold away (the host selects it only for a handle with exactly
// this configuration, kuka_tree.hip: spec_config_of).
// PERSIST = 1 (srlhip_set_persistent; GIVEN actions): the loop does not count to T — before every step the wavefront waits for the host's
// next sequence number (PersistArgs: workgroup 0 polls the mapped word, the others its relay in device memory), reads its actions from
// the SAME mapped row every step, writes its outputs straight to the host's mapped planes (they stay in its XCD's L2; the last wavefront
// of its eighth of the grid — one XCD, verified behind a start barrier — writes that L2 back and reports; on any other placement the
// outputs go through a staging copy that this wavefront copies out), publishes Monitor's record of an episode that ended right away;
// it leaves the loop (and writes the state back like any rollout) when workgroup 0 relays the park token.  While it waits it already
// runs the action-independent half of the next step (tree::tphysics_pre).
// (timeline build of persistent stepping, profiles/probes/persist_timeline.py: -DSRL_PERSIST_PROF; 100 MHz device-wide clock, the stamps of
//  a workgroup's LAST step, 8 per workgroup, behind the relay / counter words)
#if defined(SRL_PERSIST_PROF) && defined(__HIP_DEVICE_COMPILE__)
#define SRL_PSTAMP(k) do { if (threadIdx.x == 0) reinterpret_cast<uint64_t *>(pa.relay + 28 * kPersistWordStride)[bid * 8 + (k)] = wall_clock64(); } while (0)
#else
#define SRL_PSTAMP(k) do { } while (0)
#endif

// persistent stepping: one wavefront copies an env range of the three output planes staging -> mapped host planes.  The staging copy was
// written through by wavefronts of every XCD: agent-scope loads (sc1), ALL of them in flight before the first store — 16 bytes per lane
// where the range is 16-byte aligned (always for obs / reward: ranges start at a multiple of 4 envs), dwords for the rest.
struct PersistSeg { const uint32_t *src; uint32_t *dst; int count; };      // (dwords)
__device__ __forceinline__ void persist_copy(const PersistSeg (&seg)[3]) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    constexpr int kQ = 9;                                                    // 16-byte pieces in flight per lane: 9 KB per pass
    int n4[3];                                                               // 16-byte pieces of each segment (0: not aligned)
#pragma unroll
    for (int k = 0; k < 3; k++)
        n4[k] = ((reinterpret_cast<uintptr_t>(seg[k].src) | reinterpret_cast<uintptr_t>(seg[k].dst)) & 15) == 0 ? seg[k].count / 4 : 0;
    const int pieces = n4[0] + n4[1] + n4[2];
    for (int base = 0; base < pieces; base += kQ * kGroupBlock) {
        u32x4 w[kQ];
        const uint32_t *sp[kQ];
        int64_t off[kQ];
#pragma unroll
        for (int q = 0; q < kQ; q++) {
            int g = base + q * kGroupBlock + (int)threadIdx.x;
            const bool on = g < pieces;
            g = on ? g : 0;
            const int k = g < n4[0] ? 0 : g < n4[0] + n4[1] ? 1 : 2;
            const int idx = g - (k == 0 ? 0 : k == 1 ? n4[0] : n4[0] + n4[1]);
            sp[q] = seg[k].src + (int64_t)idx * 4;
            off[q] = on ? (seg[k].dst + (int64_t)idx * 4) - seg[k].src - (int64_t)idx * 4 : 0;   // dst - src of the piece's segment; 0 = skip
            // (unconditional — a lane past the end re-reads piece 0 — so that no select sits between the load and the wait below)
            asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(w[q]) : "v"(sp[q]) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int q = 0; q < kQ; q++)
            if (off[q]) *reinterpret_cast<u32x4 *>(const_cast<uint32_t *>(sp[q]) + off[q]) = w[q];
    }
    // what is left: unaligned segments whole, the last count % 4 dwords of aligned ones
#pragma unroll
    for (int k = 0; k < 3; k++)
        for (int i = n4[k] * 4 + (int)threadIdx.x; i < seg[k].count; i += kGroupBlock)
            seg[k].dst[i] = __hip_atomic_load(seg[k].src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}

template <int MODE, bool JOINTS, bool GIVEN, int NB, int RB = 0, int SPEC = 0, int PERSIST = 0>
__global__ void __launch_bounds__(kGroupBlock)
kuka_tree_rollout_k(KukaParams p, KukaState s, RngState rs, EpisodeStats st, int T, const void *actions, const double *noise,
                    float *obs, float *rew, uint8_t *done_out, void *act_out, PersistArgs pa) {
    static_assert(!PERSIST || GIVEN, "persistent stepping takes the caller's actions");
    using namespace grp;
    __shared__ double scratch_all[kGroupEnvs][kTS];
    const int64_t n = p.n;
    // XCD-aware block -> env map (the grid is a multiple of 8 blocks, kuka_tree.hip): workgroup b runs on XCD b % 8 and every XCD has its
    // own L2, so with env = 4 b + ... the 8 wavefronts whose 16-byte pieces make up one 128-byte line of an output plane sat on 8
    // different L2s and every line went to HBM in pieces (WRITE_SIZE 1.95x the planes, rounds 3-5).  XCD x now owns the contiguous env
    // range [x, x + 1) * nb / 8 * 4: a line is assembled in ONE L2.  Placement is a speed matter only; results do not depend on it.
    const int bid = ((int)blockIdx.x & 7) * ((int)gridDim.x >> 3) + ((int)blockIdx.x >> 3);
    if constexpr (PERSIST) {
#if defined(__HIP_DEVICE_COMPILE__)
        // Every workgroup (the padding ones too) registers with the XCD it runs on: the direct output path below is valid only if the
        // eighth of the grid that shares an arrival counter (blockIdx % 8) shares an L2, i.e. IS one XCD.  One word: count | mismatches << 16.
        uint32_t xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        if (threadIdx.x == 0) __hip_atomic_fetch_add(pa.ctrl, 1u + (((xcc & 15u) != (blockIdx.x & 7u)) ? 0x10000u : 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
    }
    if (bid * kGroupEnvs >= p.n) return;             // a padding block of the rounded-up grid (whole wavefront): it must not shadow env n - 1
                                                     // from another wavefront (GroupMt regenerates the env's generator state in HBM in place)
    const int e_raw = bid * kGroupEnvs + (int)(threadIdx.x / GL);
    const bool valid = e_raw < p.n;
    const int e = valid ? e_raw : p.n - 1;           // tail groups shadow the last env (every lane stays active for the cross-lane ops)
    Cfg cfg_c = p.cfg;
    if constexpr (SPEC == 1) {
        cfg_c.is_discrete = 1; cfg_c.action_joints = 0; cfg_c.random_target = 0; cfg_c.force_down = 1; cfg_c.shape_reward = 0;
        cfg_c.action_repeat = 1; cfg_c.obs_mode = 0; cfg_c.auto_reset = 1; cfg_c.moving = 0; cfg_c.two = 0; cfg_c.rand_objects = 0;
        cfg_c.max_steps = kMaxSteps;
    }
    const Cfg &cfg = cfg_c;
    __shared__ double tab[tree::kLaneTableDoubles];
    double *scratch = scratch_all[threadIdx.x / GL];
    LaneId L;
    build_lane_table(L, s.ttable, tab);
#if defined(SRL_TREE_PROF) && defined(__HIP_DEVICE_COMPILE__)
    if (threadIdx.x == 0) { unsigned long long *pp = tree::tprof_buf(); for (int i = 0; i < tree::kProfSlots; i++) pp[i] = 0; pp[tree::kProfSlots] = __builtin_amdgcn_s_memtime(); }
#endif
    const bool lead = L.l == 0 && valid;
    using Rng = std::conditional_t<MODE == SRLHIP_RNG_PHILOX, GroupPhilox, std::conditional_t<MODE == SRLHIP_RNG_MT19937, GroupMt, typename KRng<MODE>::type>>;
    Rng rng0;
    if constexpr (MODE == SRLHIP_RNG_PHILOX) rng0.init(rs.key[e], rs.key[n + e], rs.ctr[e]);
    else if constexpr (MODE == SRLHIP_RNG_MT19937) rng0.load(rs.mt, e);
    else krng_load<MODE>(rng0, rs, e, p.n, noise ? noise + e : nullptr);
    Env v = {};
    GState g;
    tload(s, n, e, L, v, g, NB == 2);
    tree::RBody body = {};
    if constexpr (RB) tload_body(s, n, e, L.l, body);
    // (requested before the forward kinematics so that their round trip overlaps it: single-step launches are latency)
    // (Monitor's record of the last finished episode is written only when an episode finishes in this launch and is not read: on
    //  host-pointer handles those two planes are mapped host memory — srlhip_episode_records — and a read / an unconditional
    //  write-back would cross PCIe in every launch)
    double ep_ret = st.ep_return[e], last_ret = 0.0, last_reward = 0.0;
    int32_t ep_len = st.ep_length[e], last_len = 0, n_fin = st.n_finished[e];
    const int32_t n_fin0 = n_fin;
    GroupActions gact; gact.init(rs.key[e], rs.key[n + e], rs.act_ctr[e]);
    tree::tfk(tree::lane_view(tab), g);
    Philox &act = gact.p;
    const int od = cfg.obs_mode == 1 ? 14 : cfg.obs_mode == 2 ? 17 : 3;
    const int adim = cfg.is_discrete ? 1 : cfg.action_joints ? 7 : 3;
    // Every load of the prologue retires HERE.  Otherwise the compiler's wait for them sits at their first use INSIDE the loop, as
    // `s_waitcnt vmcnt(0)` — and gfx9 counts stores on the same counter, so from the second step on that wait drains the previous
    // step's output stores: a full store round trip (~1.5 k cycles) in every step of the rollout.
    __builtin_amdgcn_s_waitcnt(0x0F70);
    // Per-lane RUNNING plane pointers in VGPRs, advanced by one [n]-row per step: with the base pointers left in the kernel arguments
    // the compiler, short of SGPRs, re-read them from the kernarg segment at every store site (s_load + s_waitcnt lgkmcnt(0): four
    // scalar-cache round trips per step) and rebuilt each address with 64-bit multiplies.
    float *obs_p = obs ? obs + (int64_t)e * od : nullptr, *rew_p = rew ? rew + e : nullptr;
    uint8_t *done_p = done_out ? done_out + e : nullptr;
    char *act_p = act_out ? static_cast<char *>(act_out) + (int64_t)e * adim * 4 : nullptr;
    const char *given_p = GIVEN ? static_cast<const char *>(actions) + (int64_t)e * adim * 4 : nullptr;
    const int64_t act_stride = n * adim * 4;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(obs_p), "+v"(rew_p), "+v"(done_p), "+v"(act_p), "+v"(given_p));
#endif
    uint32_t my_seq = pa.start_seq, persist_k = 0;      // persist_k: steps since this launch
    (void)my_seq; (void)persist_k;
    bool direct = false, park_now = false;
    (void)direct; (void)park_now;
    if constexpr (PERSIST) {
#if defined(__HIP_DEVICE_COMPILE__)
        // Start barrier (the grid is co-resident): workgroup 0 waits until every workgroup has registered and publishes the verdict —
        // 1: every eighth of the grid sits on one XCD -> the wavefronts write their outputs STRAIGHT to the host's mapped planes (plain
        // stores: they stay in that XCD's L2) and the eighth's last arriver writes the L2 back once; 2: not so -> the staging copy and
        // the copier below (valid on any placement); 3: told to stop while waiting (a workgroup never started) -> everybody parks.
        uint32_t verdict = 0;
        if (threadIdx.x == 0) {
            uint32_t *vw = pa.ctrl + kPersistWordStride;
            if (blockIdx.x == 0) {
                for (;;) {
                    const uint32_t reg = __hip_atomic_load(pa.ctrl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((reg & 0xffffu) == gridDim.x) { verdict = ((reg >> 16) || pa.force_staged) ? 2u : 1u; break; }
                    if (__hip_atomic_load(pa.stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) { verdict = 3u; break; }
                    __builtin_amdgcn_s_sleep(8);
                }
                __hip_atomic_store(vw, verdict, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                while (!(verdict = __hip_atomic_load(vw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) __builtin_amdgcn_s_sleep(8);
            }
        }
        verdict = __builtin_amdgcn_readfirstlane(verdict);
        direct = verdict == 1u;
        park_now = verdict == 3u;
        if (direct) {
            obs_p = reinterpret_cast<float *>(pa.host_out) + (int64_t)e * od;
            rew_p = reinterpret_cast<float *>(pa.host_out + pa.rew_dw) + e;
            done_p = reinterpret_cast<uint8_t *>(pa.host_out + pa.done_dw) + e;
        }
#endif
    }
    for (int t = 0; (PERSIST && !park_now) || (!PERSIST && t < T); t++) {
        tree::PreDyn pre;
        tree::PreStep pre2;
        // (the launching kernel of the default configuration with the caller's actions — the per-step path of HipVecEnv — takes the same
        //  split: its action is REQUESTED first, from mapped host memory on a single-step launch, and everything in front of the action
        //  runs under that PCIe round trip)
        constexpr bool kLaunchSplit = GIVEN && !PERSIST && SPEC == 1;
        int a = 0; float ca[7] = {0, 0, 0, 0, 0, 0, 0};
        if constexpr (kLaunchSplit) {
            a = *reinterpret_cast<const int32_t *>(given_p);
            given_p += act_stride;
            tree::tphysics_pre2<0>(v, g, tab, scratch, pre2);
        }
        if constexpr (PERSIST && SPEC == 1) {
            // everything of the step that depends on the state only — dynamics, collision detection, the rows up to their right-hand
            // sides and, on a contact step, the general path's whole setup (16 k cycles of the step that sets the batch's latency) —
            // runs BEFORE the wait for the host's action
            tree::tphysics_pre2<0>(v, g, tab, scratch, pre2);
        } else if constexpr (PERSIST) {
            // the action-independent half of the step's (first) physics step — joint axes, bias forces, the mass matrix and its inverse:
            // 8 k of a free step's 53 k cycles — runs BEFORE the wait for the host's action: off the step's latency
            tree::tphysics_pre(g, tab, pre);
        }
        if constexpr (PERSIST) {
#if defined(__HIP_DEVICE_COMPILE__)
            uint32_t token = 0;
            if (threadIdx.x == 0) {
                if (blockIdx.x == 0) {
                    // the one poller on the bus: a new sequence number -> relay it; told to stop, or nothing for spin_limit polls -> park
                    uint32_t sq = my_seq, stop = 0, spins = 0;
                    for (;;) {
                        // (seq, stop) are one aligned 8-byte word of the control block: ONE PCIe read per poll
                        const uint64_t w = __hip_atomic_load(reinterpret_cast<const uint64_t *>(pa.seq), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        sq = (uint32_t)w; stop = (uint32_t)(w >> 32);
                        if (sq != my_seq || stop || ++spins >= pa.spin_limit) break;
                        __builtin_amdgcn_s_sleep(8);
                    }
                    token = sq != my_seq ? sq : kPersistPark;
                    if (token == kPersistPark) __hip_atomic_store(pa.parked, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    // eight relay words, 256 bytes apart (eight memory channels): a poller reads the word of its blockIdx % 8
                    for (int x = 0; x < 8; x++) __hip_atomic_store(pa.relay + x * kPersistWordStride, token, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    const uint32_t *rw = pa.relay + (blockIdx.x & 7) * kPersistWordStride;
                    for (;;) {
                        token = __hip_atomic_load(rw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (token != my_seq) break;
                        __builtin_amdgcn_s_sleep(8);              // ~128 pollers per word, ~0.25 us between two polls of a wavefront
                    }
                }
            }
            token = __builtin_amdgcn_readfirstlane(token);
            if (token == kPersistPark) break;
            my_seq = token;
            persist_k += 1;
            SRL_PSTAMP(0);
            asm volatile("" ::: "memory");                       // the action reads below stay behind the token (they are system-scope loads)
#endif
        }
        if constexpr (GIVEN) {
            if constexpr (kLaunchSplit) {
            } else if constexpr (PERSIST) {           // the same mapped row every step: re-read, never cached in a register
                if (cfg.is_discrete) a = __hip_atomic_load(reinterpret_cast<const int32_t *>(given_p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                else for (int j = 0; j < adim; j++) ca[j] = __hip_atomic_load(reinterpret_cast<const float *>(given_p) + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            } else {
                if (cfg.is_discrete) a = *reinterpret_cast<const int32_t *>(given_p);
                else for (int j = 0; j < adim; j++) ca[j] = reinterpret_cast<const float *>(given_p)[j];
                given_p += act_stride;
            }
        } else {
            if (cfg.is_discrete) a = gact.next(5);
            else for (int j = 0; j < adim; j += 2) {
                uint32_t o[4]; act.block(o);
                ca[j] = (float)(-1.0 + 2.0 * Philox::to_double(o[0], o[1]));
                if (j + 1 < adim) ca[j + 1] = (float)(-1.0 + 2.0 * Philox::to_double(o[2], o[3]));
            }
            if (act_p && lead) {
                if (cfg.is_discrete) *reinterpret_cast<int32_t *>(act_p) = a;
                else for (int j = 0; j < adim; j++) reinterpret_cast<float *>(act_p)[j] = ca[j];
            }
            if (act_p) act_p += act_stride;
        }
        float ca_own = 0.f;
#pragma unroll
        for (int j = 0; j < ND; j++) ca_own = L.l == j ? ca[j] : ca_own;
#if defined(SRL_PERSIST_PROF) && defined(__HIP_DEVICE_COMPILE__)
        if constexpr (PERSIST) { __builtin_amdgcn_s_waitcnt(0x0F70); SRL_PSTAMP(1); }
#endif
#if defined(SRL_TREE_PROF) && defined(__HIP_DEVICE_COMPILE__)
        { using namespace tree; SRL_TSTAMP(20); }     // loop back-edge + the agent's action (sampled or loaded)
#endif
        bool done;
        double reward;
        if constexpr ((PERSIST || kLaunchSplit) && SPEC == 1) reward = tree::tenv_step<NB, RB, 0, 0, 2>(v, g, tab, cfg, scratch, rng0, a, ca, ca_own, &done, &body, nullptr, nullptr, &pre2);
        else if constexpr (PERSIST) reward = tree::tenv_step<NB, RB, 0, SPEC ? 0 : -1, 1>(v, g, tab, cfg, scratch, rng0, a, ca, ca_own, &done, &body, nullptr, &pre);
        else reward = tree::tenv_step<NB, RB, 0, SPEC ? 0 : -1>(v, g, tab, cfg, scratch, rng0, a, ca, ca_own, &done, &body);
        ep_ret += reward; ep_len += 1; last_reward = reward;
        const int info = cfg.info_bits ? (v.ikx & 1) << 1 : 0;      // srlhip_config.info_bits: the IK conditioning flag this step ran under (before the auto-reset clears it)
        if (done) {
            last_ret = ep_ret; last_len = ep_len; n_fin += 1; ep_ret = 0.0; ep_len = 0;
            if (cfg.auto_reset) {
                double *objs = valid ? s.objs + e : nullptr;
                tree::tenv_reset<JOINTS ? 1 : 0, NB, RB>(v, g, tab, cfg, scratch, rng0, s.tstarts, s.tsettled, objs, n, &body);
                // the start-state loads retire HERE, not at their first use in the next step (where vmcnt(0) would also wait for
                // the output stores of steps that did not reset)
                __builtin_amdgcn_s_waitcnt(0x0F70);
            }
        }
        if constexpr (PERSIST) {
#if defined(__HIP_DEVICE_COMPILE__)
            // The outputs go to a STAGING copy of the host's planes in device memory, by agent-scope (write-through) stores.  Neither
            // way of writing them to the mapped planes directly works from 1024 wavefronts: a plain store stays in the XCD's L2 until a
            // write-back (measured: the host saw the previous step's observations; a release fence per wavefront writes back the whole
            // L2 — generator states, spills — 1024 times per step: 121 us), a system-scope store of 1-12 bytes crosses PCIe as its own
            // serialised transaction (~40 ns each, 20 k per step: 835 us).  Monitor's record of an episode that ended in this step goes
            // to its mapped plane directly (rare: system-scope stores).
            SRL_PSTAMP(2);
            if (lead) {
                float ob[17];
                observe(v, cfg, ob, 1);
                if (direct) {                                    // plain stores to the mapped planes: into this XCD's L2
                    for (int j = 0; j < od; j++) obs_p[j] = ob[j];
                    *rew_p = (float)reward;
                    *done_p = (uint8_t)((int)done | info);
                } else {
                    for (int j = 0; j < od; j++) __hip_atomic_store(obs_p + j, ob[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(rew_p, (float)reward, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(done_p, (uint8_t)((int)done | info), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (done) {
                    __hip_atomic_store(st.last_return + e, last_ret, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    __hip_atomic_store(st.last_length + e, last_len, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
            __builtin_amdgcn_s_waitcnt(0x0F70);         // "written through" = the store counter reaching 0
            asm volatile("" ::: "memory");
            SRL_PSTAMP(3);
            // Every eighth of the workgroups (a contiguous env range) has an arrival counter; the LAST wavefront to arrive — staged
            // form: copies that range of the three planes from the staging copy to the host's, dwords, coalesced, whole lines —
            // makes them visible with ONE system-scope release (a write-back of its L2: in the direct form that IS the transfer) and
            // then writes the eighth's `done` word: the host polls 8 words.  The counter is never reset: after k steps it stands at
            // k * (real workgroups of the eighth).
            const int per = (int)gridDim.x >> 3, grp8 = bid / per;
            int real = (p.n + kGroupEnvs - 1) / kGroupEnvs - grp8 * per;
            real = real > per ? per : real;
            uint32_t last = 0;
            if (threadIdx.x == 0) last = __hip_atomic_fetch_add(pa.count + grp8 * kPersistWordStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == (uint32_t)real * persist_k;
            SRL_PSTAMP(4);
            if (__builtin_amdgcn_readfirstlane(last)) {
                const int lo = grp8 * per * kGroupEnvs, hi = min(lo + per * kGroupEnvs, p.n);
                const PersistSeg seg[3] = {{pa.stage + (int64_t)lo * od, pa.host_out + (int64_t)lo * od, (hi - lo) * od},
                                           {pa.stage + pa.rew_dw + lo, pa.host_out + pa.rew_dw + lo, hi - lo},
                                           {pa.stage + pa.done_dw + lo / 4, pa.host_out + pa.done_dw + lo / 4, (hi - lo + 3) / 4}};
                if (!direct) persist_copy(seg);
                SRL_PSTAMP(5);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
                SRL_PSTAMP(6);
                if (threadIdx.x == 0) __hip_atomic_store(pa.done + grp8, my_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
#endif
        } else {
            if (lead) {
                if (obs_p) observe(v, cfg, obs_p, 1);
                if (rew_p) *rew_p = (float)reward;
                if (done_p) *done_p = (uint8_t)((int)done | info);
            }
            if (obs_p) obs_p += n * od;
            if (rew_p) rew_p += n;
            if (done_p) done_p += n;
            if constexpr (GIVEN) {
#if defined(__HIP_DEVICE_COMPILE__)
                // EARLY COMPLETION SIGNAL of a single-step launch on a host-pointer handle (api.hip host_step_begin arms it: pa.done set):
                // the host does not wait for the kernel to END (exit stores of ~40 state planes, the completion signal, the stream
                // synchronisation's wake-up) — the step's outputs are plain stores to its mapped planes in this XCD's L2, and the last
                // wavefront of each eighth of the grid (one XCD — checked per launch, below; otherwise the host is told to wait for the
                // kernel's end) writes that L2 back and reports, as in persistent stepping.
                if (pa.done && t == T - 1) {
                    if (done && lead) {          // Monitor's record: the host reads it right after the step, before the exit stores below
                        __hip_atomic_store(st.last_return + e, last_ret, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        __hip_atomic_store(st.last_length + e, last_len, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                    // (which XCD is immaterial — a second kernel running beside this one shifts the round-robin — as long as the eighth's
                    //  workgroups all sit on the SAME one: each ORs its XCD's bit into the eighth's tag before it arrives)
                    const int per = (int)gridDim.x >> 3, grp8 = bid / per;
                    uint32_t xcc;
                    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
                    uint32_t *tag = pa.count + grp8 * kPersistWordStride + 1;
                    if (threadIdx.x == 0) __hip_atomic_fetch_or(tag, 1u << (xcc & 15u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __builtin_amdgcn_s_waitcnt(0x0F70);
                    asm volatile("" ::: "memory");
                    int real = (p.n + kGroupEnvs - 1) / kGroupEnvs - grp8 * per;
                    real = real > per ? per : real;
                    uint32_t last = 0;
                    if (threadIdx.x == 0) last = __hip_atomic_fetch_add(pa.count + grp8 * kPersistWordStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == (uint32_t)real * pa.start_seq;
                    if (__builtin_amdgcn_readfirstlane(last)) {
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
                        if (threadIdx.x == 0) {
                            const uint32_t seen = __hip_atomic_exchange(tag, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // (reset for the next launch)
                            // one XCD: the write-back above carried the whole eighth -> the sequence number; several: its complement
                            // (the host then waits for the kernel's end, where every L2 is written back)
                            __hip_atomic_store(pa.done + grp8, (seen & (seen - 1u)) ? ~pa.start_seq : pa.start_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        }
                    }
                }
#endif
            }
        }
#if defined(SRL_TREE_PROF) && defined(__HIP_DEVICE_COMPILE__)
        { using namespace tree; SRL_TSTAMP(11); }     // episode statistics, auto-reset, observation + output stores
#endif
    }
#if defined(SRL_TREE_PROF) && defined(__HIP_DEVICE_COMPILE__)
    if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == 511)) {
        const unsigned long long *pp = tree::tprof_buf();
        for (int i = 0; i < tree::kProfSlots; i++) printf("tprof block %d T %d phase %d cycles %llu\n", (int)blockIdx.x, T, i, pp[i]);
    }
#endif
    int e_out = e;
    asm volatile("" : "+v"(e_out));       // exit-store addresses are recomputed instead of being kept live across the loop
    tstore(s, n, e_out, L, v, g, valid, NB == 2);
    if constexpr (RB) tstore_body(s, n, e_out, L.l, body, valid);
    if (lead) {
        if constexpr (MODE == SRLHIP_RNG_PHILOX) rs.ctr[e_out] = rng0.p.ctr;
        else if constexpr (MODE == SRLHIP_RNG_MT19937) rng0.store(rs.mt, e_out);
        else krng_store<MODE>(rng0, rs, e_out);
        if constexpr (!GIVEN) rs.act_ctr[e_out] = act.ctr;
        st.ep_return[e_out] = ep_ret; st.ep_length[e_out] = ep_len;
        if (n_fin != n_fin0) { st.last_return[e_out] = last_ret; st.last_length[e_out] = last_len; }
        st.n_finished[e_out] = n_fin; st.last_reward[e_out] = last_reward;
    }
}

// srlhip_reset: KukaButtonGymEnv.reset by lane groups
template <int MODE, bool JOINTS, int NB, int RB = 0>
__global__ void __launch_bounds__(kGroupBlock)
kuka_tree_reset_k(KukaParams p, KukaState s, RngState rs, EpisodeStats st, const uint8_t *mask, const double *host_rand, int rand_stride, float *obs) {
    using namespace grp;
    __shared__ double scratch_all[kGroupEnvs][kTS];
    const int64_t n = p.n;
    const int e_raw = blockIdx.x * kGroupEnvs + (int)(threadIdx.x / GL);
    const bool valid = e_raw < p.n && !(mask && !mask[e_raw < p.n ? e_raw : 0]);
    const int e = e_raw < p.n ? e_raw : p.n - 1;
    __shared__ double tab[tree::kLaneTableDoubles];
    LaneId L; build_lane_table(L, s.ttable, tab);
    const bool lead = L.l == 0 && valid;
    using Rng = std::conditional_t<MODE == SRLHIP_RNG_PHILOX, GroupPhilox, std::conditional_t<MODE == SRLHIP_RNG_MT19937, GroupMt, typename KRng<MODE>::type>>;
    // A masked-out env (and the shadow rows of a tail group) must not draw: GroupMt's twist() rewrites the env's 624 state words in
    // HBM in place while its index is only stored by the lead lane of a VALID env — a masked draw across a twist boundary would
    // leave the running env with a regenerated block and a stale index.  `valid` is uniform over the 16-lane row, and every
    // cross-lane operation of the reset is row-local (the rollout kernel already runs tenv_reset under row divergence).
    if (!valid) return;
    Rng rng0;
    if constexpr (MODE == SRLHIP_RNG_PHILOX) rng0.init(rs.key[e], rs.key[n + e], rs.ctr[e]);
    else if constexpr (MODE == SRLHIP_RNG_MT19937) rng0.load(rs.mt, e);
    else krng_load<MODE>(rng0, rs, e, p.n, host_rand ? host_rand + (int64_t)e * rand_stride : nullptr);
    Env v = {};
    v.ikx = s.i[I_IKX * n + e];              // the flagged-step count outlives the episode (tenv_reset clears the sticky bit only)
    GState g;
    double *objs = s.objs + e;
    tree::RBody body = {};
    tree::tenv_reset<JOINTS ? 1 : 0, NB, RB>(v, g, tab, p.cfg, scratch_all[threadIdx.x / GL], rng0, s.tstarts, s.tsettled, objs, n, &body);
    tstore(s, n, e, L, v, g, valid, NB == 2);
    if constexpr (RB) tstore_body(s, n, e, L.l, body, valid);
    if (lead) {
        if constexpr (MODE == SRLHIP_RNG_PHILOX) rs.ctr[e] = rng0.p.ctr;
        else if constexpr (MODE == SRLHIP_RNG_MT19937) rng0.store(rs.mt, e);
        else krng_store<MODE>(rng0, rs, e);
        st.ep_return[e] = 0.0; st.ep_length[e] = 0;
        if (obs) {
            const int od = p.cfg.obs_mode == 1 ? 14 : p.cfg.obs_mode == 2 ? 17 : 3;
            observe(v, p.cfg, obs + (int64_t)e * od, 1);
        }
    }
}

}  // namespace
}  // namespace srl
