// kuka_hostcheck.cpp — TESTS ONLY.  Compiles the kernel's own physics source
// (kuka_core.hpp / kuka_env.hpp, all __host__ __device__) for the host so the
// CPU-side suite (-m "not gpu") can check the HIP stepper's arithmetic against
// the oracle without a GPU.  It is a separate shared object
// (csrc/build/libsrlhip_hostcheck.so) that the product package never loads and
// libsrlhip.so does not contain: there is no CPU fallback in the product.
#include <stdlib.h>
#include <string.h>

#include <vector>

// row-count instrumentation: ngen (limit + contact rows) of every env step, for the path statistics in DESIGN.md
static unsigned char *g_stat = nullptr; static long g_stat_idx = -1;
#define SRL_ROW_STAT_HOOK(ngen, nlim) do { if (g_stat && g_stat_idx >= 0) g_stat[g_stat_idx] = (unsigned char)((ngen) | ((nlim) > 0 ? 0x80 : 0)); } while (0)
#include "kuka_env.hpp"

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
        double *f = final_state + 30 * (size_t)e_idx;
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
    Cfg cfg;
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
