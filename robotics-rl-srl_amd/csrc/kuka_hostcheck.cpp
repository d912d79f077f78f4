// kuka_hostcheck.cpp — TESTS ONLY.  Compiles the kernel's own physics source
// (kuka_core.hpp / kuka_env.hpp, all __host__ __device__) for the host so the
// CPU-side suite (-m "not gpu") can check the HIP stepper's arithmetic against
// the oracle without a GPU.  It is a separate shared object
// (csrc/build/libsrlhip_hostcheck.so) that the product package never loads and
// libsrlhip.so does not contain: there is no CPU fallback in the product.
#include <stdlib.h>
#include <string.h>

#include <limits>
#include <vector>

// row-count instrumentation: ngen (limit + contact rows) of every env step, for the path statistics in DESIGN.md
static unsigned char *g_stat = nullptr; static long g_stat_idx = -1;
#define SRL_ROW_STAT_HOOK(ngen, nlim) do { if (g_stat && g_stat_idx >= 0) g_stat[g_stat_idx] = (unsigned char)((ngen) | ((nlim) > 0 ? 0x80 : 0)); } while (0)
#include "kuka_env.hpp"
static double g_gdbg[8][128]; static int g_gdbg_on = 0;
#define SRL_GDBG(tag, idx, val) do { if (g_gdbg_on && (idx) < 128) g_gdbg[tag][idx] = (val); } while (0)
static long g_fix_hist[152];
#define SRL_GDBG_FIXPOINT(at) do { if (g_gdbg_on) g_fix_hist[(at) < 0 ? 151 : (at)]++; } while (0)
static long g_clamp_cnt[2];
#define SRL_GDBG_CLAMPS(cl) do { if (g_gdbg_on) { g_clamp_cnt[0]++; g_clamp_cnt[1] += (cl); } } while (0)
#define SRL_GDBG_COUNTS(any, gl, gc, gb) do { if (g_gdbg_on) { g_gdbg[6][0] += 1; g_gdbg[6][1] += (any); g_gdbg[6][2] += (gl); g_gdbg[6][3] += (gc); g_gdbg[6][4] += (gb); } } while (0)
#include "kuka_group.hpp"
#include "kuka_tree.hpp"

using namespace srl;
using namespace srl::kuka;

namespace {
int g_moving = 0, g_two = 0, g_rand = 0;
struct MtHost {     // host-side generator over a private SoA view with stride 1
    std::vector<uint32_t> words;
    int32_t mti, has_g; double g;
    Mt19937 m;
    MtHost() : words(MT_N) {}
    void seed(const uint32_t *key, int len) {
        Mt19937View v{words.data(), &mti, &has_g, &g, 1};
        m.load(v, 0);
        m.seed_by_array(key, len);
    }
    double double01() { return m.double01(); }
    double uniform(double a, double b) { return m.uniform(a, b); }
    double normal(double a, double b) { return m.normal(a, b); }
    uint32_t bounded(uint32_t r) { return m.bounded(r); }
};
struct PhHost {
    Philox p;
    double double01() { return p.double01(); }
    double uniform(double a, double b) { return p.uniform(a, b); }
    double normal(double a, double b) { return p.normal(a, b); }
    uint32_t bounded(uint32_t r) { return p.bounded(r); }
};

void build_tables(const Cfg &cfg, std::vector<double> &settled, std::vector<double> &starts) {
    std::vector<double> scratch(SC_TOTAL), rows(SC_ROWS_TOTAL);
    Scratch sc{scratch.data(), 1, rows.data(), 1, nullptr};
    Env e;
    initial_env(e);
    const double zero[3] = {0, 0, 0};
    double jt[ND];
    for (int j = 0; j < ND; j++) jt[j] = kJointPositions[j];
    for (int i = 0; i < kNSettleSteps; i++) physics_step<1>(e, cfg, sc, zero, cfg.action_joints != 0, jt);
    settled.resize(kStartDoubles);
    pack_start(e, settled.data());
    if (!cfg.is_discrete && cfg.action_joints) return;
    const int nstarts = cfg.is_discrete ? kNumStartsDiscrete : kNumStartsContinuous, base = cfg.is_discrete ? 6 : 2;
    starts.resize((size_t)nstarts * kStartDoubles);
    for (int idx = 0; idx < nstarts; idx++) {
        Env s = e;
        int rem = idx;
        for (int k = 0; k < kNInitActions; k++) {
            double motor[3];
            init_action_motor(cfg, rem % base, motor);
            physics_step<1>(s, cfg, sc, motor, false, jt);
            rem /= base;
        }
        pack_start(s, starts.data() + (size_t)idx * kStartDoubles);
    }
}

template <int NB, class R>
void run_env(const Cfg &cfg, R &rng, Philox act, int T, int n, int e_idx, const void *actions, const double *settled,
             const double *starts, float *obs0, float *obs, float *rew, double *rew64, uint8_t *done_out, void *act_out,
             double *q_trace, double *grip_trace, double *final_state, double *ep_stats) {
    std::vector<double> scratch(SC_TOTAL), rows(SC_ROWS_TOTAL);
    Scratch sc{scratch.data(), 1, rows.data(), 1, nullptr};
    const int od = cfg.obs_mode == 1 ? 14 : cfg.obs_mode == 2 ? 17 : 3;
    const int adim = cfg.is_discrete ? 1 : cfg.action_joints ? 7 : 3;
    Env env;
    memset(&env, 0, sizeof env);
    reset_env<NB>(env, cfg, sc, rng, starts, settled);
    if (obs0) observe(env, cfg, obs0 + (size_t)e_idx * od, 1);
    double ep_ret = 0, last_ret = 0; int ep_len = 0, last_len = 0, n_fin = 0;
    for (int t = 0; t < T; t++) {
        const size_t row = (size_t)t * n + e_idx;
        int a = 0; float ca[7] = {0}; bool done;
        if (actions) {
            if (cfg.is_discrete) a = static_cast<const int32_t *>(actions)[row];
            else memcpy(ca, static_cast<const float *>(actions) + row * adim, sizeof(float) * adim);
        } else {
            if (cfg.is_discrete) a = (int)act.bounded(5);
            else for (int j = 0; j < adim; j += 2) {
                uint32_t o[4]; act.block(o);
                ca[j] = (float)(-1.0 + 2.0 * Philox::to_double(o[0], o[1]));
                if (j + 1 < adim) ca[j + 1] = (float)(-1.0 + 2.0 * Philox::to_double(o[2], o[3]));
            }
            if (act_out) { if (cfg.is_discrete) static_cast<int32_t *>(act_out)[row] = a; else memcpy(static_cast<float *>(act_out) + row * adim, ca, sizeof(float) * adim); }
        }
        g_stat_idx = (long)row;
        const double reward = env_step<NB>(env, cfg, sc, rng, a, ca, &done);
        g_stat_idx = -1;
        if (q_trace) memcpy(q_trace + row * ND, env.q, sizeof(double) * ND);
        if (grip_trace) memcpy(grip_trace + row * 3, env.grip, sizeof(double) * 3);
        ep_ret += reward; ep_len += 1;
        if (done) {
            last_ret = ep_ret; last_len = ep_len; n_fin += 1; ep_ret = 0; ep_len = 0;
            if (cfg.auto_reset) reset_env<NB>(env, cfg, sc, rng, starts, settled);
        }
        if (obs) observe(env, cfg, obs + row * od, 1);
        if (rew) rew[row] = (float)reward;
        if (rew64) rew64[row] = reward;
        if (done_out) done_out[row] = (uint8_t)done;
    }
    if (final_state) {
        double *f = final_state + 40 * (size_t)e_idx;   // stride of oracle.kuka_clib.rollout (columns 30..39: gripper DoFs of the full model)
        for (int j = 0; j < ND; j++) { f[j] = env.q[j]; f[7 + j] = env.qd[j]; }
        f[14] = env.ee[0]; f[15] = env.ee[1]; f[16] = env.ee[2]; f[17] = env.bq; f[18] = env.bqd; f[19] = env.counter;
        f[20] = env.n_contacts; f[21] = env.n_outside; f[22] = env.terminated; f[23] = cfg.moving ? env.bpos[1] : env.bpos[2];
        if (NB == 2) { f[24] = env.b2q; f[25] = env.b2qd; f[26] = env.goal_id; f[27] = env.n_contacts2; f[28] = env.b2x; f[29] = env.b2y; }
    }
    if (ep_stats) { ep_stats[3 * (size_t)e_idx] = last_ret; ep_stats[3 * (size_t)e_idx + 1] = last_len; ep_stats[3 * (size_t)e_idx + 2] = n_fin; }
}
}  // namespace

// same signature as oracle/kuka_oracle.c:kuka_oracle_rollout so one test drives both
extern "C" int hostcheck_kuka_rollout(int is_discrete, int action_joints, int random_target, int force_down,
                                      int shape_reward, int action_repeat, double max_distance, int obs_mode,
                                      int rng_mode, int auto_reset, int n, int T, const int64_t *seeds,
                                      const uint32_t *mt_keys, const int32_t *mt_key_len, const void *actions,
                                      float *obs0, float *obs, float *rew, double *rew64, uint8_t *done_out,
                                      void *act_out, double *q_trace, double *grip_trace, double *final_state,
                                      double *ep_stats) {
    Cfg cfg; memset(&cfg, 0, sizeof cfg);
    cfg.random_target = random_target; cfg.force_down = force_down; cfg.shape_reward = shape_reward;
    cfg.action_repeat = action_repeat; cfg.is_discrete = is_discrete; cfg.action_joints = action_joints;
    cfg.obs_mode = obs_mode; cfg.auto_reset = auto_reset; cfg.max_distance = max_distance;
    cfg.moving = g_moving; cfg.two = g_two; cfg.max_steps = g_moving ? 1500 : g_two ? kMaxSteps2Button : kMaxSteps;
    cfg.rand_objects = g_rand;
    std::vector<double> settled, starts;
    build_tables(cfg, settled, starts);
    for (int e = 0; e < n; e++) {
        Philox act; act.k0 = (uint32_t)(uint64_t)seeds[e]; act.k1 = (uint32_t)((uint64_t)seeds[e] >> 32); act.ctr = 0; act.stream = 1;
        if (rng_mode == 2) {
            MtHost r; r.seed(mt_keys + 2 * (size_t)e, mt_key_len[e]);
            if (g_two) run_env<2>(cfg, r, act, T, n, e, actions, settled.data(), starts.data(), obs0, obs, rew, rew64, done_out, act_out, q_trace, grip_trace, final_state, ep_stats);
            else run_env<1>(cfg, r, act, T, n, e, actions, settled.data(), starts.data(), obs0, obs, rew, rew64, done_out, act_out, q_trace, grip_trace, final_state, ep_stats);
        } else {
            PhHost r; r.p = act; r.p.stream = 0;
            if (g_two) run_env<2>(cfg, r, act, T, n, e, actions, settled.data(), starts.data(), obs0, obs, rew, rew64, done_out, act_out, q_trace, grip_trace, final_state, ep_stats);
            else run_env<1>(cfg, r, act, T, n, e, actions, settled.data(), starts.data(), obs0, obs, rew, rew64, done_out, act_out, q_trace, grip_trace, final_state, ep_stats);
        }
    }
    return 0;
}

extern "C" void hostcheck_kuka_set_row_stats(unsigned char *buf) { g_stat = buf; }
extern "C" void hostcheck_kuka_set_moving(int m) { g_moving = m; g_two = 0; g_rand = 0; }
extern "C" void hostcheck_kuka_set_variant(int v) { g_moving = v == 1; g_two = v == 2; g_rand = v == 3; }

extern "C" void hostcheck_kuka_settled(int random_target, int action_joints, double *out22) {
    Cfg cfg; memset(&cfg, 0, sizeof cfg);
    cfg.random_target = random_target; cfg.action_joints = action_joints; cfg.action_repeat = 1; cfg.is_discrete = 1;
    std::vector<double> settled, starts;
    cfg.is_discrete = 0; cfg.action_joints = 1;        // skip the start table
    cfg.action_joints = action_joints ? 1 : 1;
    {   // settle only
        std::vector<double> scratch(SC_TOTAL), rows(SC_ROWS_TOTAL);
        Scratch sc{scratch.data(), 1, rows.data(), 1, nullptr};
        Env e; initial_env(e);
        const double zero[3] = {0, 0, 0}; double jt[ND];
        for (int j = 0; j < ND; j++) jt[j] = kJointPositions[j];
        cfg.is_discrete = 1; cfg.action_joints = action_joints; cfg.moving = 0; cfg.two = g_two; cfg.rand_objects = 0; cfg.max_steps = kMaxSteps;
        for (int i = 0; i < kNSettleSteps; i++) physics_step<1>(e, cfg, sc, zero, action_joints != 0, jt);
        for (int j = 0; j < ND; j++) { out22[j] = e.q[j]; out22[7 + j] = e.qd[j]; }
        out22[14] = e.ee[0]; out22[15] = e.ee[1]; out22[16] = e.ee[2]; out22[17] = e.bq; out22[18] = e.bqd;
        out22[19] = e.grip[0]; out22[20] = e.grip[1]; out22[21] = e.grip[2];
    }
}


// =====================================================================================================================
// Lane-group stepper (kuka_group.hpp) on the host: the 16 lanes of an env's DPP row are 16 cooperatively scheduled fibers
// that run the kernel's own SIMT source in lockstep; every cross-lane primitive is one value exchange (write own slot,
// yield round-robin through the other 15 fibers, read the source lane's slot).
namespace {
extern "C" void srl_fiber_switch(void **save_sp, void *new_sp);
}
asm(R"(
    .text
    .globl srl_fiber_switch
    .type srl_fiber_switch,@function
srl_fiber_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
)");

namespace {
struct FiberGroup {
    static constexpr int N = grp::GL;
    static constexpr size_t kStack = 1 << 20;
    void *sp[N], *main_sp;
    std::vector<char> stacks;
    int cur = 0, done = 0;
    long ops[N];
    double slot[2][N];
    uint32_t vote[2];
    void (*body)(void *);
    void *arg;
    FiberGroup() : stacks(N * kStack) {}
};
FiberGroup *g_group = nullptr;

void fiber_yield() {
    FiberGroup *G = g_group;
    const int me = G->cur;
    G->cur = (me + 1) % FiberGroup::N;
    srl_fiber_switch(&G->sp[me], G->sp[G->cur]);
}
void fiber_entry() {
    FiberGroup *G = g_group;
    G->body(G->arg);
    // lockstep: fibers finish in lane order; hand over to the next one (still inside its last exchange), the last to main
    const int me = G->cur;
    G->done++;
    if (me + 1 < FiberGroup::N) { G->cur = me + 1; srl_fiber_switch(&G->sp[me], G->sp[me + 1]); }
    else srl_fiber_switch(&G->sp[me], G->main_sp);
    abort();
}
void run_group(void (*body)(void *), void *arg) {
    static FiberGroup group;
    FiberGroup *G = &group;
    g_group = G;
    G->body = body; G->arg = arg; G->cur = 0; G->done = 0; G->vote[0] = G->vote[1] = 0;
    for (int l = 0; l < FiberGroup::N; l++) {
        G->ops[l] = 0;
        char *top = G->stacks.data() + (size_t)(l + 1) * FiberGroup::kStack;
        uintptr_t a = ((uintptr_t)top - 64) & ~(uintptr_t)15;          // 16-aligned: after 6 pops + ret, rsp = a + 56 == 8 (mod 16)
        void **st = (void **)a;
        for (int k = 0; k < 6; k++) st[k] = nullptr;
        st[6] = (void *)&fiber_entry;
        st[7] = nullptr;
        G->sp[l] = st;
    }
    srl_fiber_switch(&G->main_sp, G->sp[0]);
    if (G->done != FiberGroup::N) abort();
    for (int l = 1; l < FiberGroup::N; l++) if (G->ops[l] != G->ops[0]) abort();      // every lane ran the same cross-lane ops
}
}  // namespace

namespace srl { namespace kuka { namespace grp {
int host_lane() { return g_group->cur; }
double host_exchange(double x, int src) {
    FiberGroup *G = g_group;
    const int me = G->cur, buf = (int)(G->ops[me]++ & 1);
    G->slot[buf][me] = x;
    fiber_yield();
    return G->slot[buf][src];
}
uint32_t host_ballot(bool p) {
    FiberGroup *G = g_group;
    const int me = G->cur, buf = (int)(G->ops[me]++ & 1);
    G->slot[buf][me] = p ? 1.0 : 0.0;
    fiber_yield();
    uint32_t m = 0;
    for (int l = 0; l < FiberGroup::N; l++) if (G->slot[buf][l] != 0.0) m |= 1u << l;
    return m;
}
}}}  // namespace srl::kuka::grp

namespace {
Model g_model; bool g_model_set = false;
struct GroupArgs {
    const Model *model;        // runtime model table, or null for the baked one
    Cfg cfg; int rng_mode; const uint32_t *mt_key; int mt_key_len; Philox act; int T, n, e_idx; const void *actions;
    const double *settled, *starts; float *obs0, *obs, *rew; double *rew64; uint8_t *done_out; void *act_out;
    double *q_trace, *grip_trace, *final_state, *ep_stats;
    MtHost *mt; double *scratch;
};

template <bool CM, class R>
void group_env_body(GroupArgs &a, R &rng) {
    using namespace grp;
    const Cfg &cfg = a.cfg;
    const int n = a.n, e_idx = a.e_idx, T = a.T;
    const int od = cfg.obs_mode == 1 ? 14 : cfg.obs_mode == 2 ? 17 : 3;
    const int adim = cfg.is_discrete ? 1 : cfg.action_joints ? 7 : 3;
    Lane L; lane_init<CM>(L, a.model);
    const bool lead = L.l == 0;
    Env env; memset(&env, 0, sizeof env);
    GState g; memset(&g, 0, sizeof g);
    const bool joints = !cfg.is_discrete && cfg.action_joints;
    if (joints) genv_reset<true, CM>(env, g, L, cfg, a.scratch, rng, a.starts, a.settled, nullptr, 1);
    else genv_reset<false, CM>(env, g, L, cfg, a.scratch, rng, a.starts, a.settled, nullptr, 1);
    if (a.obs0 && lead) observe(env, cfg, a.obs0 + (size_t)e_idx * od, 1);
    Philox act = a.act;
    GroupActions gact; gact.init(a.act.k0, a.act.k1, 0);
    double ep_ret = 0, last_ret = 0; int ep_len = 0, last_len = 0, n_fin = 0;
    for (int t = 0; t < T; t++) {
        const size_t row = (size_t)t * n + e_idx;
        int ac = 0; float ca[7] = {0}; bool done;
        if (a.actions) {
            if (cfg.is_discrete) ac = static_cast<const int32_t *>(a.actions)[row];
            else memcpy(ca, static_cast<const float *>(a.actions) + row * adim, sizeof(float) * adim);
        } else {
            if (cfg.is_discrete) ac = gact.next(5);
            else for (int j = 0; j < adim; j += 2) {
                uint32_t o[4]; act.block(o);
                ca[j] = (float)(-1.0 + 2.0 * Philox::to_double(o[0], o[1]));
                if (j + 1 < adim) ca[j + 1] = (float)(-1.0 + 2.0 * Philox::to_double(o[2], o[3]));
            }
            if (a.act_out && lead) { if (cfg.is_discrete) static_cast<int32_t *>(a.act_out)[row] = ac; else memcpy(static_cast<float *>(a.act_out) + row * adim, ca, sizeof(float) * adim); }
        }
        const double reward = genv_step<CM>(env, g, L, cfg, a.scratch, rng, ac, ca, L.arm ? ca[L.l] : 0.f, &done);
        if (a.q_trace && L.arm) a.q_trace[row * ND + L.l] = g.q;
        if (a.grip_trace && lead) memcpy(a.grip_trace + row * 3, env.grip, sizeof(double) * 3);
        ep_ret += reward; ep_len += 1;
        if (done) {
            last_ret = ep_ret; last_len = ep_len; n_fin += 1; ep_ret = 0; ep_len = 0;
            if (cfg.auto_reset) {
                if (joints) genv_reset<true, CM>(env, g, L, cfg, a.scratch, rng, a.starts, a.settled, nullptr, 1);
                else genv_reset<false, CM>(env, g, L, cfg, a.scratch, rng, a.starts, a.settled, nullptr, 1);
            }
        }
        if (lead) {
            if (a.obs) observe(env, cfg, a.obs + row * od, 1);
            if (a.rew) a.rew[row] = (float)reward;
            if (a.rew64) a.rew64[row] = reward;
            if (a.done_out) a.done_out[row] = (uint8_t)done;
        }
    }
    if (a.final_state) {
        double *f = a.final_state + 40 * (size_t)e_idx;
        if (L.arm) { f[L.l] = g.q; f[7 + L.l] = g.qd; }
        if (lead) {
            f[14] = env.ee[0]; f[15] = env.ee[1]; f[16] = env.ee[2]; f[17] = env.bq; f[18] = env.bqd; f[19] = env.counter;
            f[20] = env.n_contacts; f[21] = env.n_outside; f[22] = env.terminated; f[23] = cfg.moving ? env.bpos[1] : env.bpos[2];
        }
    }
    if (a.ep_stats && lead) { a.ep_stats[3 * (size_t)e_idx] = last_ret; a.ep_stats[3 * (size_t)e_idx + 1] = last_len; a.ep_stats[3 * (size_t)e_idx + 2] = n_fin; }
}

void group_fiber_body(void *p) {
    GroupArgs &a = *static_cast<GroupArgs *>(p);
    if (a.rng_mode == 2) {
        grp::Lane0Rng<MtHost> r{a.mt, grp::lane_id() == 0};
        if (a.model) group_env_body<true>(a, r); else group_env_body<false>(a, r);
    } else {
        grp::GroupPhilox r; r.init(a.act.k0, a.act.k1, 0);      // counter-based: every lane holds the same stream
        if (a.model) group_env_body<true>(a, r); else group_env_body<false>(a, r);
    }
}

// settled state of a runtime model table: 500 zero-action steps of one lane group (kuka_button_gym_env.py:242-247)
struct SettleArgs { const Model *model; Cfg cfg; double *out36; double *scratch; };
void group_settle_body(void *p) {
    using namespace grp;
    SettleArgs &a = *static_cast<SettleArgs *>(p);
    Lane L; lane_init<true>(L, a.model);
    Env e; memset(&e, 0, sizeof e);
    GState g; memset(&g, 0, sizeof g);
    g.q = L.arm ? L.q0 : 0.0; g.qd = 0.0;
    for (int k = 0; k < 3; k++) e.ee[k] = kEeInit[k];
    e.bx = kButtonX; e.by = kButtonY; e.bz = L.base_z;
    grefresh<true>(L, g, e);
    const double zero[3] = {0, 0, 0};
    for (int i = 0; i < kNSettleSteps; i++) gphysics_step<true>(e, g, L, a.cfg, a.scratch, zero, a.cfg.action_joints != 0, L.q0);
    double *o = a.out36;
    if (L.arm) { o[L.l] = g.q; o[7 + L.l] = g.qd; o[14 + L.l] = g.sq; o[21 + L.l] = g.cq; }
    if (L.l == 0) { for (int k = 0; k < 3; k++) { o[28 + k] = e.ee[k]; o[33 + k] = e.grip[k]; } o[31] = e.bq; o[32] = e.bqd; }
}
}  // namespace

// same signature as hostcheck_kuka_rollout; KukaButton / Moving / RandButton (the lane-group kernel has no two-button form)
extern "C" int hostcheck_kuka_group_rollout(int is_discrete, int action_joints, int random_target, int force_down,
                                            int shape_reward, int action_repeat, double max_distance, int obs_mode,
                                            int rng_mode, int auto_reset, int n, int T, const int64_t *seeds,
                                            const uint32_t *mt_keys, const int32_t *mt_key_len, const void *actions,
                                            float *obs0, float *obs, float *rew, double *rew64, uint8_t *done_out,
                                            void *act_out, double *q_trace, double *grip_trace, double *final_state,
                                            double *ep_stats) {
    if (g_two) return -1;
    GroupArgs a;
    a.model = g_model_set ? &g_model : nullptr;
    Cfg &cfg = a.cfg;
    cfg.random_target = random_target; cfg.force_down = force_down; cfg.shape_reward = shape_reward;
    cfg.action_repeat = action_repeat; cfg.is_discrete = is_discrete; cfg.action_joints = action_joints;
    cfg.obs_mode = obs_mode; cfg.auto_reset = auto_reset; cfg.max_distance = max_distance;
    cfg.moving = g_moving; cfg.two = 0; cfg.max_steps = g_moving ? 1500 : kMaxSteps; cfg.rand_objects = g_rand;
    std::vector<double> settled, starts, scratch(grp::kScratchDoubles);
    if (a.model) {            // runtime table: settle with the lane-group stepper itself; no start table (resets integrate their init actions)
        settled.assign(kStartDoubles, 0.0);
        SettleArgs sa{a.model, cfg, settled.data(), scratch.data()};
        run_group(group_settle_body, &sa);
    } else build_tables(cfg, settled, starts);
    a.rng_mode = rng_mode; a.T = T; a.n = n; a.actions = actions; a.settled = settled.data(); a.starts = starts.data();
    a.obs0 = obs0; a.obs = obs; a.rew = rew; a.rew64 = rew64; a.done_out = done_out; a.act_out = act_out;
    a.q_trace = q_trace; a.grip_trace = grip_trace; a.final_state = final_state; a.ep_stats = ep_stats; a.scratch = scratch.data();
    for (int e = 0; e < n; e++) {
        a.e_idx = e;
        a.act.k0 = (uint32_t)(uint64_t)seeds[e]; a.act.k1 = (uint32_t)((uint64_t)seeds[e] >> 32); a.act.ctr = 0; a.act.stream = 1;
        MtHost mt;
        if (rng_mode == 2) mt.seed(mt_keys + 2 * (size_t)e, mt_key_len[e]);
        a.mt = &mt;
        run_group(group_fiber_body, &a);
    }
    return 0;
}

extern "C" void hostcheck_group_debug(int on, double *out) { g_gdbg_on = on; if (out) memcpy(out, g_gdbg, sizeof g_gdbg); }

extern "C" void hostcheck_group_fix_hist(long *out) { memcpy(out, g_fix_hist, sizeof g_fix_hist); memset(g_fix_hist, 0, sizeof g_fix_hist); }

extern "C" void hostcheck_group_clamp_cnt(long *out) { out[0] = g_clamp_cnt[0]; out[1] = g_clamp_cnt[1]; g_clamp_cnt[0] = g_clamp_cnt[1] = 0; }

// runtime model table for the lane-group harness (138 doubles, srlhip_kuka_model layout); null -> the baked model
extern "C" void hostcheck_kuka_set_model(const double *table138) {
    g_model_set = table138 != nullptr;
    if (table138) memcpy(&g_model, table138, sizeof g_model);
}
extern "C" void hostcheck_kuka_default_model(double *table138) { Model m; default_model(m); memcpy(table138, &m, sizeof m); }


// =====================================================================================================================
// Full-model lane-group stepper (kuka_tree.hpp) under the same fiber harness.  Resets integrate their init actions from the
// settled state (START = 2 / 1): the start-state table is a device-side shortcut for the same arithmetic.
namespace {
srl::kuka::TreeModel g_tree_model; bool g_tree_model_set = false;
const srl::kuka::TreeModel *tree_model() {
    if (!g_tree_model_set) { srl::kuka::default_tree_model(g_tree_model); g_tree_model_set = true; }
    return &g_tree_model;
}
double *g_tree_bodies = nullptr;         // KukaRandButton: [n][11][7] body state at the end of the next tree rollout (x y z vx vy vz on)
uint8_t *g_tree_ik_flag = nullptr; int32_t *g_tree_ik_fin = nullptr;     // IK conditioning flag: [T][n] sticky bit after each step, [n] final Env::ikx
int g_tree_occ = 0;                      // 1: the two-wavefronts-per-SIMD variant's code path (OCC = 1: shared work area, per-env park, recomputed candidates)
template <int NB, int RB, int OCC, class R>
void tree_env_body(GroupArgs &a, R &rng) {
    using namespace tree;
    const Cfg &cfg = a.cfg;
    const int n = a.n, e_idx = a.e_idx, T = a.T;
    const int od = cfg.obs_mode == 1 ? 14 : cfg.obs_mode == 2 ? 17 : 3;
    const int adim = cfg.is_discrete ? 1 : cfg.action_joints ? 7 : 3;
    TLane L; lane_init(L, tree_model());
    static double tab[kLaneTableDoubles];          // the wavefront's lane-constant table (LDS on the device): shared by the 16 fibers
    lane_store(L, tab);
    grp::sync_scratch();
    const bool lead = L.l == 0;
    Env env; memset(&env, 0, sizeof env);
    GState g; memset(&g, 0, sizeof g);
    RBody body; memset(&body, 0, sizeof body);
    static double park[kTreeParkDoubles];          // OCC: the env's park area (shared by the 16 fibers)
    if (L.l == 0) for (int i = 0; i < kTreeParkDoubles; i++) park[i] = std::numeric_limits<double>::quiet_NaN();
    grp::sync_scratch();
    const bool joints = !cfg.is_discrete && cfg.action_joints;
    if (joints) tenv_reset<1, NB, RB, OCC>(env, g, tab, cfg, a.scratch, rng, a.starts, a.settled, nullptr, 1, &body, park);
    else tenv_reset<2, NB, RB, OCC>(env, g, tab, cfg, a.scratch, rng, a.starts, a.settled, nullptr, 1, &body, park);
    if (a.obs0 && lead) observe(env, cfg, a.obs0 + (size_t)e_idx * od, 1);
    Philox act = a.act;
    grp::GroupActions gact; gact.init(a.act.k0, a.act.k1, 0);
    double ep_ret = 0, last_ret = 0; int ep_len = 0, last_len = 0, n_fin = 0;
    for (int t = 0; t < T; t++) {
        const size_t row = (size_t)t * n + e_idx;
        int ac = 0; float ca[7] = {0}; bool done;
        if (a.actions) {
            if (cfg.is_discrete) ac = static_cast<const int32_t *>(a.actions)[row];
            else memcpy(ca, static_cast<const float *>(a.actions) + row * adim, sizeof(float) * adim);
        } else {
            if (cfg.is_discrete) ac = gact.next(5);
            else for (int j = 0; j < adim; j += 2) {
                uint32_t o[4]; act.block(o);
                ca[j] = (float)(-1.0 + 2.0 * Philox::to_double(o[0], o[1]));
                if (j + 1 < adim) ca[j + 1] = (float)(-1.0 + 2.0 * Philox::to_double(o[2], o[3]));
            }
            if (a.act_out && lead) { if (cfg.is_discrete) static_cast<int32_t *>(a.act_out)[row] = ac; else memcpy(static_cast<float *>(a.act_out) + row * adim, ca, sizeof(float) * adim); }
        }
        const double reward = tenv_step<NB, RB, OCC>(env, g, tab, cfg, a.scratch, rng, ac, ca, L.arm ? ca[L.l] : 0.f, &done, &body, park);
        if (a.q_trace && L.arm) a.q_trace[row * ND + L.l] = g.q;
        if (a.grip_trace && lead) memcpy(a.grip_trace + row * 3, env.grip, sizeof(double) * 3);
        if (g_tree_ik_flag && lead) g_tree_ik_flag[row] = (uint8_t)(env.ikx & 1);
        ep_ret += reward; ep_len += 1;
        if (done) {
            last_ret = ep_ret; last_len = ep_len; n_fin += 1; ep_ret = 0; ep_len = 0;
            if (cfg.auto_reset) {
                if (joints) tenv_reset<1, NB, RB, OCC>(env, g, tab, cfg, a.scratch, rng, a.starts, a.settled, nullptr, 1, &body, park);
                else tenv_reset<2, NB, RB, OCC>(env, g, tab, cfg, a.scratch, rng, a.starts, a.settled, nullptr, 1, &body, park);
            }
        }
        if (lead) {
            if (a.obs) observe(env, cfg, a.obs + row * od, 1);
            if (a.rew) a.rew[row] = (float)reward;
            if (a.rew64) a.rew64[row] = reward;
            if (a.done_out) a.done_out[row] = (uint8_t)done;
        }
    }
    if (a.final_state) {
        double *f = a.final_state + 40 * (size_t)e_idx;
        if (L.arm) { f[L.l] = g.q; f[7 + L.l] = g.qd; }
        else if (L.jnt) { f[30 + L.l - ND] = g.q; f[35 + L.l - ND] = g.qd; }
        if (lead) {
            f[14] = env.ee[0]; f[15] = env.ee[1]; f[16] = env.ee[2]; f[17] = env.bq; f[18] = env.bqd; f[19] = env.counter;
            f[20] = env.n_contacts; f[21] = env.n_outside; f[22] = env.terminated; f[23] = cfg.moving ? env.bpos[1] : env.bpos[2];
            if (NB == 2) { f[24] = env.b2q; f[25] = env.b2qd; f[26] = env.goal_id; f[27] = env.n_contacts2; f[28] = env.b2x; f[29] = env.b2y; }
        }
    }
    if (a.ep_stats && lead) { a.ep_stats[3 * (size_t)e_idx] = last_ret; a.ep_stats[3 * (size_t)e_idx + 1] = last_len; a.ep_stats[3 * (size_t)e_idx + 2] = n_fin; }
    if (g_tree_ik_fin && lead) g_tree_ik_fin[e_idx] = env.ikx;
    if (RB && g_tree_bodies && L.l < kRbN) {
        double *bd = g_tree_bodies + ((size_t)e_idx * kRbN + L.l) * 7;
        for (int k = 0; k < 3; k++) { bd[k] = body.x[k]; bd[3 + k] = body.v[k]; }
        bd[6] = body.on ? 1.0 : 0.0;
    }
}
template <class R> void tree_env_dispatch(GroupArgs &a, R &r) {
    if (a.cfg.two) tree_env_body<2, 0, 0>(a, r);
    else if (a.cfg.rand_objects) tree_env_body<1, 1, 0>(a, r);
    else if (g_tree_occ) tree_env_body<1, 0, 1>(a, r);
    else tree_env_body<1, 0, 0>(a, r);
}
void tree_fiber_body(void *p) {
    GroupArgs &a = *static_cast<GroupArgs *>(p);
    if (a.rng_mode == 2) {            // the lane-group MT19937 of the device kernels over the env's host-side state words (every fiber: its own replica)
        grp::GroupMt r; r.m = a.mt->m; r.win = 0u; r.pos = 0; r.cnt = 0;
        tree_env_dispatch(a, r);
    }
    else { grp::GroupPhilox r; r.init(a.act.k0, a.act.k1, 0); tree_env_dispatch(a, r); }
}
struct TreeSettleArgs { Cfg cfg; double *out; double *scratch; };
void tree_settle_body(void *p) {
    using namespace tree;
    TreeSettleArgs &a = *static_cast<TreeSettleArgs *>(p);
    TLane L; lane_init(L, tree_model());
    static double tab[kLaneTableDoubles];
    lane_store(L, tab);
    grp::sync_scratch();
    Env e; memset(&e, 0, sizeof e);
    GState g; memset(&g, 0, sizeof g);
    tinitial(e, g, tab);
    const double zero[3] = {0, 0, 0};
    for (int i = 0; i < kNSettleSteps; i++) tphysics_step(e, g, tab, a.cfg, a.scratch, zero, a.cfg.action_joints != 0, L.q0, 0.0);
    tpack_start(e, g, a.out);
}
}  // namespace

extern "C" int hostcheck_kuka_tree_rollout(int is_discrete, int action_joints, int random_target, int force_down,
                                           int shape_reward, int action_repeat, double max_distance, int obs_mode,
                                           int rng_mode, int auto_reset, int n, int T, const int64_t *seeds,
                                           const uint32_t *mt_keys, const int32_t *mt_key_len, const void *actions,
                                           float *obs0, float *obs, float *rew, double *rew64, uint8_t *done_out,
                                           void *act_out, double *q_trace, double *grip_trace, double *final_state,
                                           double *ep_stats) {
    GroupArgs a;
    a.model = nullptr;
    Cfg &cfg = a.cfg;
    cfg.random_target = random_target; cfg.force_down = force_down; cfg.shape_reward = shape_reward;
    cfg.action_repeat = action_repeat; cfg.is_discrete = is_discrete; cfg.action_joints = action_joints;
    cfg.obs_mode = obs_mode; cfg.auto_reset = auto_reset; cfg.max_distance = max_distance;
    cfg.moving = g_moving; cfg.two = g_two; cfg.max_steps = g_moving ? 1500 : g_two ? kMaxSteps2Button : kMaxSteps; cfg.rand_objects = g_rand;
    // LDS is not cleared between launches on the device: poison the emulated scratch so that any read of a slot this env never
    // wrote (0 * stale = NaN) shows up here and not as a flaky GPU test
    std::vector<double> settled(tree::kTreeStartDoubles, 0.0), starts, scratch(tree::kTreeScratchDoubles, std::numeric_limits<double>::quiet_NaN());
    TreeSettleArgs sa{cfg, settled.data(), scratch.data()};
    run_group(tree_settle_body, &sa);
    a.rng_mode = rng_mode; a.T = T; a.n = n; a.actions = actions; a.settled = settled.data(); a.starts = starts.data();
    a.obs0 = obs0; a.obs = obs; a.rew = rew; a.rew64 = rew64; a.done_out = done_out; a.act_out = act_out;
    a.q_trace = q_trace; a.grip_trace = grip_trace; a.final_state = final_state; a.ep_stats = ep_stats; a.scratch = scratch.data();
    for (int e = 0; e < n; e++) {
        a.e_idx = e;
        a.act.k0 = (uint32_t)(uint64_t)seeds[e]; a.act.k1 = (uint32_t)((uint64_t)seeds[e] >> 32); a.act.ctr = 0; a.act.stream = 1;
        MtHost mt;
        if (rng_mode == 2) mt.seed(mt_keys + 2 * (size_t)e, mt_key_len[e]);
        a.mt = &mt;
        run_group(tree_fiber_body, &a);
    }
    return 0;
}
extern "C" void hostcheck_kuka_tree_set_body_trace(double *bodies) { g_tree_bodies = bodies; }
extern "C" void hostcheck_kuka_tree_set_ik_trace(uint8_t *flag, int32_t *fin) { g_tree_ik_flag = flag; g_tree_ik_fin = fin; }
extern "C" void hostcheck_kuka_tree_set_occ(int occ) { g_tree_occ = occ; }
extern "C" void hostcheck_kuka_tree_default_model(double *t510) { srl::kuka::TreeModel m; srl::kuka::default_tree_model(m); memcpy(t510, &m, sizeof m); }
extern "C" void hostcheck_kuka_tree_set_model(const double *t510) {
    g_tree_model_set = t510 != nullptr;
    if (t510) memcpy(&g_tree_model, t510, sizeof g_tree_model);
}
