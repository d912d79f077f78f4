// kuka.hip — KukaButtonGymEnv stepper kernels for gfx950 (MI355X) and their host plumbing.
//
// Launch geometry: one lane per env, 64-lane workgroups (one wavefront), per-link ABA
// staging + the first two constraint rows in LDS ([slot][lane], 70 KiB per workgroup -> 2 per CU), further rows in an
// L2-resident scratch,
// state structure-of-arrays in HBM (env index fastest: a wavefront's loads and
// stores of a field are one coalesced 512-byte row).  kuka_rollout_k keeps the
// whole env state in VGPRs for T steps and streams the [T][N] observation /
// reward / done planes.  The path is FP64-VALU / dependency-latency bound (150
// sequential Gauss-Seidel sweeps per physics step), not HBM bound and not
// MFMA-shaped; see DESIGN.md §Kuka kernel for the roofline accounting.
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "kuka_device.hpp"
#include "kuka_tree.hpp"

namespace srl {
using namespace kuka;

namespace {

template <int NB>
__device__ __forceinline__ void load_env(const KukaState &s, int64_t n, int64_t e, Env &v) {
    if constexpr (NB == 2) {
        v.b2q = s.d[D_B2Q * n + e]; v.b2qd = s.d[D_B2QD * n + e]; v.b2x = s.d[D_B2X * n + e]; v.b2y = s.d[D_B2Y * n + e];
        v.goal_id = s.i[I_GOAL * n + e]; v.n_contacts2 = s.i[I_NCONTACT2 * n + e]; v.contact_body1 = 0; v.contact_body2 = 0;
    }
#pragma unroll
    for (int k = 0; k < ND; k++) {
        v.q[k] = s.d[(D_Q + k) * n + e]; v.qd[k] = s.d[(D_QD + k) * n + e];
        v.sq[k] = s.d[(D_SQ + k) * n + e]; v.cq[k] = s.d[(D_CQ + k) * n + e];
    }
#pragma unroll
    for (int k = 0; k < 3; k++) { v.ee[k] = s.d[(D_EE + k) * n + e]; v.bpos[k] = s.d[(D_BPOS + k) * n + e]; v.grip[k] = s.d[(D_GRIP + k) * n + e]; }
    v.bq = s.d[D_BQ * n + e]; v.bqd = s.d[D_BQD * n + e]; v.bx = s.d[D_BX * n + e]; v.by = s.d[D_BY * n + e];
    v.bz = s.d[D_BZ * n + e]; v.bspeed = s.d[D_BSPEED * n + e];
    v.motor_on = s.i[I_MOTOR * n + e]; v.contact_button = s.i[I_CB * n + e]; v.contact_table = s.i[I_CT * n + e];
    v.counter = s.i[I_COUNTER * n + e]; v.n_contacts = s.i[I_NCONTACT * n + e]; v.n_outside = s.i[I_NOUT * n + e];
    v.terminated = s.i[I_TERM * n + e];
}
template <int NB>
__device__ __forceinline__ void store_env(const KukaState &s, int64_t n, int64_t e, const Env &v) {
    if constexpr (NB == 2) {
        s.d[D_B2Q * n + e] = v.b2q; s.d[D_B2QD * n + e] = v.b2qd; s.d[D_B2X * n + e] = v.b2x; s.d[D_B2Y * n + e] = v.b2y;
        s.i[I_GOAL * n + e] = v.goal_id; s.i[I_NCONTACT2 * n + e] = v.n_contacts2;
    }
#pragma unroll
    for (int k = 0; k < ND; k++) {
        s.d[(D_Q + k) * n + e] = v.q[k]; s.d[(D_QD + k) * n + e] = v.qd[k];
        s.d[(D_SQ + k) * n + e] = v.sq[k]; s.d[(D_CQ + k) * n + e] = v.cq[k];
    }
#pragma unroll
    for (int k = 0; k < 3; k++) { s.d[(D_EE + k) * n + e] = v.ee[k]; s.d[(D_BPOS + k) * n + e] = v.bpos[k]; s.d[(D_GRIP + k) * n + e] = v.grip[k]; }
    s.d[D_BQ * n + e] = v.bq; s.d[D_BQD * n + e] = v.bqd; s.d[D_BX * n + e] = v.bx; s.d[D_BY * n + e] = v.by;
    s.d[D_BZ * n + e] = v.bz; s.d[D_BSPEED * n + e] = v.bspeed;
    s.i[I_MOTOR * n + e] = v.motor_on; s.i[I_CB * n + e] = v.contact_button; s.i[I_CT * n + e] = v.contact_table;
    s.i[I_COUNTER * n + e] = v.counter; s.i[I_NCONTACT * n + e] = v.n_contacts; s.i[I_NOUT * n + e] = v.n_outside;
    s.i[I_TERM * n + e] = v.terminated;
}

extern __shared__ double kuka_lds[];

__device__ __forceinline__ Scratch make_scratch(const KukaState &s, int64_t n, int64_t e) {
    Scratch sc;
    sc.b = kuka_lds + threadIdx.x; sc.st = kWave; sc.g = s.rows + e; sc.gst = n; sc.objs = s.objs + e;
    return sc;
}

// 500 settle steps (kuka_button_gym_env.py:242-247).  Every lane integrates the same env so that
// wave-level votes stay uniform; lane 0 publishes.
__global__ void __launch_bounds__(kWave) kuka_settle_k(KukaParams p, KukaState s) {
    Scratch sc = make_scratch(s, p.n, threadIdx.x % p.n);
    Env e = {};
    initial_env(e);
    const double zero[3] = {0, 0, 0};
    double jt[ND];
#pragma unroll
    for (int j = 0; j < ND; j++) jt[j] = kJointPositions[j];
    for (int i = 0; i < kNSettleSteps; i++) physics_step<1>(e, p.cfg, sc, zero, p.cfg.action_joints != 0, jt);
    if (threadIdx.x == 0) pack_start(e, s.settled);
}

// table of the 6^5 (2^5) possible episode start states
__global__ void __launch_bounds__(kWave) kuka_starts_k(KukaParams p, KukaState s) {
    const int idx = blockIdx.x * kWave + threadIdx.x;
    if (idx >= s.nstarts) return;
    Scratch sc = make_scratch(s, p.n, idx % p.n);     // start states are limit- and contact-free: rows are never touched
    Env e = {};
    unpack_start(e, s.settled);
    e.bx = kButtonX; e.by = kButtonY; e.bz = kButtonBaseZ; e.bspeed = 0.0; e.motor_on = 0; e.contact_button = 0; e.contact_table = 0;
    e.counter = 0; e.n_contacts = 0; e.n_outside = 0; e.terminated = 0;
    e.bpos[0] = e.bpos[1] = e.bpos[2] = 0.0;
    const int base = p.cfg.is_discrete ? 6 : 2;
    int rem = idx;
    // motor / jt are declared outside the loop on purpose (see the note in kuka_env.hpp:reset_env)
    double motor[3], jt[ND];
#pragma unroll
    for (int j = 0; j < ND; j++) jt[j] = kJointPositions[j];
    for (int k = 0; k < kNInitActions; k++) {
        init_action_motor(p.cfg, rem % base, motor);
        physics_step<1>(e, p.cfg, sc, motor, false, jt);
        rem /= base;
    }
    pack_start(e, s.starts + (int64_t)idx * kStartDoubles);
}

template <int MODE, int NB>
__global__ void __launch_bounds__(kWave)
kuka_reset_k(KukaParams p, KukaState s, RngState rs, EpisodeStats st, const uint8_t *mask, const double *host_rand,
             int rand_stride, float *obs) {
    const int e = blockIdx.x * kWave + threadIdx.x;
    if (e >= p.n) return;
    if (mask && !mask[e]) return;
    const int64_t n = p.n;
    Scratch sc = make_scratch(s, n, e);
    typename KRng<MODE>::type rng;
    krng_load<MODE>(rng, rs, e, p.n, host_rand ? host_rand + (int64_t)e * rand_stride : nullptr);
    Env v = {};
    reset_env<NB>(v, p.cfg, sc, rng, s.starts, s.settled);
    store_env<NB>(s, n, e, v);
    krng_store<MODE>(rng, rs, e);
    st.ep_return[e] = 0.0; st.ep_length[e] = 0;
    if (obs) {
        const int od = p.cfg.obs_mode == 1 ? 14 : p.cfg.obs_mode == 2 ? 17 : 3;
        observe(v, p.cfg, obs + (int64_t)e * od, 1);
    }
}

// T consecutive VecEnv steps per launch (T == 1: the per-step entry point).
template <int MODE, int NB>
__global__ void __launch_bounds__(kWave)
kuka_rollout_k(KukaParams p, KukaState s, RngState rs, EpisodeStats st, int T, const void *actions, const double *noise,
               float *obs, float *rew, uint8_t *done_out, void *act_out) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= p.n) return;
    const int64_t n = p.n;
    const Cfg &cfg = p.cfg;
    Scratch sc = make_scratch(s, n, e);
    typename KRng<MODE>::type rng;
    krng_load<MODE>(rng, rs, e, p.n, noise ? noise + e : nullptr);
    Env v = {};
    load_env<NB>(s, n, e, v);
    double ep_ret = st.ep_return[e], last_ret = 0.0, last_reward = 0.0;      // (last_* : written only when an episode finished, never read — mobile.hip)
    int32_t ep_len = st.ep_length[e], last_len = 0, n_fin = st.n_finished[e];
    const int32_t n_fin0 = n_fin;
    Philox act; act.k0 = rs.key[e]; act.k1 = rs.key[n + e]; act.ctr = rs.act_ctr[e]; act.stream = 1;
    const int od = cfg.obs_mode == 1 ? 14 : cfg.obs_mode == 2 ? 17 : 3;
    const int adim = cfg.is_discrete ? 1 : cfg.action_joints ? 7 : 3;
    for (int t = 0; t < T; t++) {
        const int64_t row = (int64_t)t * n + e;
        int a = 0; float ca[7] = {0, 0, 0, 0, 0, 0, 0};
        if (actions) {
            if (cfg.is_discrete) a = static_cast<const int32_t *>(actions)[row];
            else for (int j = 0; j < adim; j++) ca[j] = static_cast<const float *>(actions)[row * adim + j];
        } else {
            if (cfg.is_discrete) a = (int)act.bounded(5);
            else for (int j = 0; j < adim; j += 2) {
                uint32_t o[4]; act.block(o);
                ca[j] = (float)(-1.0 + 2.0 * Philox::to_double(o[0], o[1]));
                if (j + 1 < adim) ca[j + 1] = (float)(-1.0 + 2.0 * Philox::to_double(o[2], o[3]));
            }
            if (act_out) {
                if (cfg.is_discrete) static_cast<int32_t *>(act_out)[row] = a;
                else for (int j = 0; j < adim; j++) static_cast<float *>(act_out)[row * adim + j] = ca[j];
            }
        }
        bool done;
        const double reward = env_step<NB>(v, cfg, sc, rng, a, ca, &done);
        ep_ret += reward; ep_len += 1; last_reward = reward;
        if (done) {
            last_ret = ep_ret; last_len = ep_len; n_fin += 1; ep_ret = 0.0; ep_len = 0;
            if (cfg.auto_reset) reset_env<NB>(v, cfg, sc, rng, s.starts, s.settled);
        }
        if (obs) observe(v, cfg, obs + row * od, 1);
        if (rew) rew[row] = (float)reward;
        if (done_out) done_out[row] = (uint8_t)done;
    }
    store_env<NB>(s, n, e, v);
    krng_store<MODE>(rng, rs, e);
    if (!actions) rs.act_ctr[e] = act.ctr;
    st.ep_return[e] = ep_ret; st.ep_length[e] = ep_len;
    if (n_fin != n_fin0) { st.last_return[e] = last_ret; st.last_length[e] = last_len; }
    st.n_finished[e] = n_fin; st.last_reward[e] = last_reward;
}


// after srlhip_set_state(KUKA_Q): refresh the cached sin/cos and the gripper position
__global__ void kuka_refresh_k(KukaState s, int n) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    Env v = {};
    load_env<1>(s, n, e, v);
    update_trig_and_gripper(v);
    store_env<1>(s, n, e, v);
}


constexpr size_t kLdsBytes = (size_t)SC_TOTAL * kWave * sizeof(double);

template <class K>
int allow_lds(Handle *h, K kernel) {
    SRL_HIP_CHECK(h, hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         160 * 1024));
    return 0;
}

}  // namespace

int kuka_reset_rand_count(const srlhip_config &c) {
    const int init = c.is_discrete ? 10 : 5;
    if (c.env_kind == SRLHIP_ENV_KUKA_2BUTTON) return (c.random_target ? 4 : 0) + 2 + init;   // kuka_2button_gym_env.py:55-70
    return (c.env_kind == SRLHIP_ENV_KUKA_MOVING ? 1 : 0) + (c.random_target ? 2 : 0) + (c.env_kind == SRLHIP_ENV_KUKA_RAND ? 20 : 0) + init;
}

int kuka_alloc(Handle *h) {
    KukaState *s = new KukaState();
    h->kuka = s;
    const size_t n = (size_t)h->n;
    int rc;
    s->nstarts = (!h->cfg.is_discrete && h->cfg.action_joints) ? 0 : h->cfg.is_discrete ? kNumStartsDiscrete : kNumStartsContinuous;
    if ((rc = h->dalloc(&s->d, NDBL * n)) || (rc = h->dalloc(&s->i, NINT * n)) || (rc = h->dalloc(&s->rows, SC_ROWS_TOTAL * n)) || (rc = h->dalloc(&s->objs, 30 * n)) || (rc = h->dalloc(&s->rb, 66 * n)) ||
        (rc = h->dalloc(&s->settled, kStartDoubles)) || (rc = h->dalloc(&s->starts, (size_t)(s->nstarts > 0 ? s->nstarts : 1) * kStartDoubles)))
        return rc;
    if ((rc = h->dalloc(&s->model, 1))) return rc;
    {
        Model m; default_model(m);
        SRL_HIP_CHECK(h, hipMemcpyAsync(s->model, &m, sizeof m, hipMemcpyHostToDevice, h->stream));
        SRL_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    }
    s->custom_model = 0;
    s->full = h->cfg.kuka_model == SRLHIP_KUKA_MODEL_FULL ? 1 : 0;
    s->tmodel = nullptr; s->ttable = nullptr; s->tsettled = nullptr; s->tstarts = nullptr;
    if (s->full) {
        // the full model has its own table, settled state and start-state table; the lane-per-env kernels below are not used
        if ((rc = h->dalloc(&s->tmodel, 1)) || (rc = h->dalloc(&s->ttable, tree::kLaneTableDoubles)) || (rc = h->dalloc(&s->tsettled, tree::kTreeStartDoubles)) ||
            (rc = h->dalloc(&s->tstarts, (size_t)(s->nstarts > 0 ? s->nstarts : 1) * tree::kTreeStartDoubles)))
            return rc;
        TreeModel tm; default_tree_model(tm);
        static_assert(sizeof(TreeModel) == sizeof(h->kuka_tmodel_host), "host copy of the tree model table");
        memcpy(h->kuka_tmodel_host, &tm, sizeof tm);
        SRL_HIP_CHECK(h, hipMemcpyAsync(s->tmodel, &tm, sizeof tm, hipMemcpyHostToDevice, h->stream));
        SRL_HIP_CHECK(h, hipStreamSynchronize(h->stream));
        return kuka_tree_settle(h, params_of(h));
    }
    if ((rc = allow_lds(h, kuka_settle_k)) || (rc = allow_lds(h, kuka_starts_k))) return rc;
#define SRL_ALLOW(NB)                                                                                                     \
    if ((rc = allow_lds(h, kuka_reset_k<SRLHIP_RNG_HOST, NB>)) || (rc = allow_lds(h, kuka_reset_k<SRLHIP_RNG_PHILOX, NB>)) ||     \
        (rc = allow_lds(h, kuka_reset_k<SRLHIP_RNG_MT19937, NB>)) || (rc = allow_lds(h, kuka_rollout_k<SRLHIP_RNG_HOST, NB>)) ||  \
        (rc = allow_lds(h, kuka_rollout_k<SRLHIP_RNG_PHILOX, NB>)) || (rc = allow_lds(h, kuka_rollout_k<SRLHIP_RNG_MT19937, NB>))) \
        return rc;
    if (h->cfg.env_kind == SRLHIP_ENV_KUKA_2BUTTON) { SRL_ALLOW(2) } else { SRL_ALLOW(1) }
#undef SRL_ALLOW
    KukaParams p = params_of(h);
    hipLaunchKernelGGL(kuka_settle_k, dim3(1), dim3(kWave), kLdsBytes, h->stream, p, *s);
    SRL_HIP_CHECK(h, hipGetLastError());
    if (s->nstarts > 0) {
        hipLaunchKernelGGL(kuka_starts_k, dim3((s->nstarts + kWave - 1) / kWave), dim3(kWave), kLdsBytes, h->stream, p, *s);
        SRL_HIP_CHECK(h, hipGetLastError());
    }
    return 0;
}

void kuka_free(Handle *h) { delete h->kuka; h->kuka = nullptr; }

int kuka_reset(Handle *h, const uint8_t *d_mask, const double *d_host_rand, void *d_obs) {
    KukaParams p = params_of(h);
    dim3 grid((h->n + kWave - 1) / kWave), block(kWave);
    const int stride = kuka_reset_rand_count(h->cfg);
    float *obs = static_cast<float *>(d_obs);
    if (h->kuka->full) {
        if (h->cfg.rng_mode == SRLHIP_RNG_HOST && !d_host_rand) return h->fail(SRLHIP_EINVAL, "reset: RNG_HOST needs host_rand");
        return kuka_tree_reset(h, p, d_mask, d_host_rand, stride, obs);
    }
    if (h->kuka->custom_model) {
        if (h->cfg.rng_mode == SRLHIP_RNG_HOST && !d_host_rand) return h->fail(SRLHIP_EINVAL, "reset: RNG_HOST needs host_rand");
        return kuka_group_reset_table(h, p, d_mask, d_host_rand, stride, obs);
    }
    const bool two = h->cfg.env_kind == SRLHIP_ENV_KUKA_2BUTTON;
#define SRL_RESET(MODE)                                                                                                          \
    if (two) hipLaunchKernelGGL((kuka_reset_k<MODE, 2>), grid, block, kLdsBytes, h->stream, p, *h->kuka, h->rng, h->stats, d_mask, d_host_rand, stride, obs); \
    else hipLaunchKernelGGL((kuka_reset_k<MODE, 1>), grid, block, kLdsBytes, h->stream, p, *h->kuka, h->rng, h->stats, d_mask, d_host_rand, stride, obs);
    switch (h->cfg.rng_mode) {
        case SRLHIP_RNG_HOST:
            if (!d_host_rand) return h->fail(SRLHIP_EINVAL, "reset: RNG_HOST needs host_rand");
            SRL_RESET(SRLHIP_RNG_HOST)
            break;
        case SRLHIP_RNG_PHILOX:
            SRL_RESET(SRLHIP_RNG_PHILOX)
            break;
        default:
            SRL_RESET(SRLHIP_RNG_MT19937)
    }
#undef SRL_RESET
    SRL_HIP_CHECK(h, hipGetLastError());
    return 0;
}

// Which kernel steps a batch: the lane-group kernel fills the chip at small batches (16 lanes per env), the lane-per-env
// kernel does less total work per env once every SIMD has several wavefronts anyway.  SRLHIP_KUKA_KERNEL=group|lane forces one.
static bool use_group_kernel(const Handle *h) {
    if (h->kuka->full) return true;                                      // the tree lane-group kernel, every batch size
    if (h->cfg.env_kind == SRLHIP_ENV_KUKA_2BUTTON) return false;        // the lane-group kernel has no two-button form
    if (h->kuka->custom_model) return true;                              // only the lane-group kernel reads the runtime model table (the override below cannot undo that)
    const char *v = getenv("SRLHIP_KUKA_KERNEL");                        // read per call: tests and probes flip it inside one process
    if (v && (v[0] == 'g' || v[0] == 'l')) return v[0] == 'g';
    return h->n <= kGroupKernelMaxEnvs;
}

// srlhip_set_kuka_model: install a runtime model table; the settled state is re-integrated with it.  Envs must be reset afterwards.
int kuka_set_model(Handle *h, const double *table138) {
    if (h->kuka->full) return h->fail(SRLHIP_EINVAL, "set_kuka_model: this handle integrates the full model (srlhip_set_kuka_tree_model)");
    if (h->cfg.env_kind == SRLHIP_ENV_KUKA_2BUTTON) return h->fail(SRLHIP_ENOTSUP, "set_kuka_model: Kuka2ButtonGymEnv is stepped by the lane-per-env kernel, which is specialised for the baked model");
    KukaState *s = h->kuka;
    SRL_HIP_CHECK(h, hipMemcpyAsync(s->model, table138, sizeof(Model), hipMemcpyHostToDevice, h->stream));
    SRL_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    s->custom_model = 1;
    KukaParams p = params_of(h);
    return kuka_group_settle_table(h, p);
}
void kuka_default_model(double *table138) { Model m; default_model(m); memcpy(table138, &m, sizeof m); }

int kuka_uses_group_kernel(const Handle *h) { return h->kuka->full ? 2 : use_group_kernel(h) ? 1 : 0; }

// srlhip_set_kuka_tree_model: install a full-model table; settled state and start table are re-integrated.  Envs must be reset afterwards.
int kuka_set_tree_model(Handle *h, const double *table510) {
    KukaState *s = h->kuka;
    memcpy(h->kuka_tmodel_host, table510, sizeof(TreeModel));
    SRL_HIP_CHECK(h, hipMemcpyAsync(s->tmodel, table510, sizeof(TreeModel), hipMemcpyHostToDevice, h->stream));
    SRL_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    return kuka_tree_settle(h, params_of(h));
}
void kuka_default_tree_model(double *table510) { TreeModel m; default_tree_model(m); memcpy(table510, &m, sizeof m); }

static int kuka_group_launch(Handle *h, const KukaParams &p, int T, const void *d_actions, const double *d_noise, float *obs, float *d_rew,
                             uint8_t *d_done, void *d_act_out) {
    if (h->kuka->full) return kuka_tree_launch(h, p, T, d_actions, d_noise, obs, d_rew, d_done, d_act_out);
    return h->kuka->custom_model ? kuka_group_launch_table(h, p, T, d_actions, d_noise, obs, d_rew, d_done, d_act_out)
                                 : kuka_group_launch_baked(h, p, T, d_actions, d_noise, obs, d_rew, d_done, d_act_out);
}

int kuka_rollout(Handle *h, int T, const void *d_actions, void *d_obs, float *d_rew, uint8_t *d_done, void *d_act_out) {
    KukaParams p = params_of(h);
    if (use_group_kernel(h)) {
        if (h->cfg.rng_mode != SRLHIP_RNG_PHILOX && h->cfg.rng_mode != SRLHIP_RNG_MT19937)
            return h->fail(SRLHIP_EINVAL, "rollout: needs a device RNG mode (PHILOX or MT19937)");
        return kuka_group_launch(h, p, T, d_actions, nullptr, static_cast<float *>(d_obs), d_rew, d_done, d_act_out);
    }
    // envs per wavefront: a wavefront takes the contact path of the solver as soon as ONE of its lanes carries a contact
    // row, so fewer (active) lanes per wavefront mean fewer slow sweeps — as long as there are idle SIMDs to host the
    // extra wavefronts (experiment knob SRLHIP_KUKA_LANES; the default is chosen from the batch size below)
    static const int forced_lanes = [] { const char *v = getenv("SRLHIP_KUKA_LANES"); return v ? atoi(v) : 0; }();
    int lanes = forced_lanes == 8 || forced_lanes == 16 || forced_lanes == 32 || forced_lanes == 64 ? forced_lanes : kWave;
    dim3 grid((h->n + lanes - 1) / lanes), block(lanes);
    // LDS request of the rollout launch (experiment knob SRLHIP_KUKA_LDS_KB): > 80 KiB keeps a CU to ONE workgroup
    static const int forced_lds_kb = [] { const char *v = getenv("SRLHIP_KUKA_LDS_KB"); return v ? atoi(v) : 0; }();
    const size_t lds_bytes = forced_lds_kb > 0 && (size_t)forced_lds_kb * 1024 >= kLdsBytes && forced_lds_kb <= 160 ? (size_t)forced_lds_kb * 1024 : kLdsBytes;
    float *obs = static_cast<float *>(d_obs);
    const bool two = h->cfg.env_kind == SRLHIP_ENV_KUKA_2BUTTON;
#define SRL_ROLLOUT(MODE)                                                                                                        \
    if (two) hipLaunchKernelGGL((kuka_rollout_k<MODE, 2>), grid, block, lds_bytes, h->stream, p, *h->kuka, h->rng, h->stats, T, d_actions, (const double *)nullptr, obs, d_rew, d_done, d_act_out); \
    else hipLaunchKernelGGL((kuka_rollout_k<MODE, 1>), grid, block, lds_bytes, h->stream, p, *h->kuka, h->rng, h->stats, T, d_actions, (const double *)nullptr, obs, d_rew, d_done, d_act_out);
    switch (h->cfg.rng_mode) {
        case SRLHIP_RNG_PHILOX:
            SRL_ROLLOUT(SRLHIP_RNG_PHILOX)
            break;
        case SRLHIP_RNG_MT19937:
            SRL_ROLLOUT(SRLHIP_RNG_MT19937)
            break;
        default:
            return h->fail(SRLHIP_EINVAL, "rollout: needs a device RNG mode (PHILOX or MT19937)");
    }
#undef SRL_ROLLOUT
    SRL_HIP_CHECK(h, hipGetLastError());
    return 0;
}

int kuka_step(Handle *h, const void *d_actions, const double *d_noise, void *d_obs, float *d_rew, uint8_t *d_done) {
    if (h->cfg.rng_mode != SRLHIP_RNG_HOST) return kuka_rollout(h, 1, d_actions, d_obs, d_rew, d_done, nullptr);
    KukaParams p = params_of(h);
    if (use_group_kernel(h)) return kuka_group_launch(h, p, 1, d_actions, d_noise, static_cast<float *>(d_obs), d_rew, d_done, nullptr);
    dim3 grid((h->n + kWave - 1) / kWave), block(kWave);
    if (h->cfg.env_kind == SRLHIP_ENV_KUKA_2BUTTON)
        hipLaunchKernelGGL((kuka_rollout_k<SRLHIP_RNG_HOST, 2>), grid, block, kLdsBytes, h->stream, p, *h->kuka, h->rng, h->stats, 1,
                           d_actions, d_noise, static_cast<float *>(d_obs), d_rew, d_done, (void *)nullptr);
    else
        hipLaunchKernelGGL((kuka_rollout_k<SRLHIP_RNG_HOST, 1>), grid, block, kLdsBytes, h->stream, p, *h->kuka, h->rng, h->stats, 1,
                           d_actions, d_noise, static_cast<float *>(d_obs), d_rew, d_done, (void *)nullptr);
    SRL_HIP_CHECK(h, hipGetLastError());
    return 0;
}

int kuka_persist_blocks(Handle *h, int *capacity) { if (capacity) *capacity = 0; return h->kuka && h->kuka->full ? kuka_tree_persist_blocks(h, capacity) : 0; }
int kuka_persist_start(Handle *h, const void *d_actions, float *d_obs, float *d_rew, uint8_t *d_done, const PersistArgs &pa) {
    return kuka_tree_persist_launch(h, params_of(h), d_actions, d_obs, d_rew, d_done, pa);
}

// (RasterKukaView: internal.hpp — one definition for this file and raster.hip)
void kuka_raster_view(Handle *h, RasterKukaView *v) {
    const KukaState *s = h->kuka;
    const size_t n = (size_t)h->n;
    v->sq = s->d + D_SQ * n; v->cq = s->d + D_CQ * n; v->bq = s->d + D_BQ * n; v->bx = s->d + D_BX * n; v->by = s->d + D_BY * n; v->bz = s->d + D_BZ * n;
    v->b2q = s->d + D_B2Q * n; v->b2x = s->d + D_B2X * n; v->b2y = s->d + D_B2Y * n;
    v->n = (int64_t)n; v->two = h->cfg.env_kind == SRLHIP_ENV_KUKA_2BUTTON ? 1 : 0;
    v->objs = s->objs; v->rand_objects = h->cfg.env_kind == SRLHIP_ENV_KUKA_RAND ? 1 : 0;
    v->rb = (v->rand_objects && s->full) ? s->rb : nullptr;      // full model: the distractors and the ball are free bodies, drawn where they are
    // full model: the gripper is drawn from its own joints (gripper_to_arm, fingers, tips) through the installed table
    v->grip = nullptr; v->has_tm = s->full ? 1 : 0; v->gsq = s->d + D_GSQ * n; v->gcq = s->d + D_GCQ * n;
    for (int i = 0; i < 5; i++) {
        const TreeJoint &J = reinterpret_cast<const TreeModel *>(h->kuka_tmodel_host)->j[7 + i];
        RasterGripJoint &g = v->gj[i];
        g.parent = J.parent;
        for (int k = 0; k < 3; k++) { g.xyz[k] = J.xyz[k]; g.axis[k] = J.axis[k]; }
        for (int k = 0; k < 9; k++) g.Rj[k] = J.Rj[k];
    }
}

int kuka_refresh(Handle *h) {
    if (h->kuka->full) {
        int rc = kuka_tree_refresh(h, params_of(h));
        if (rc) return rc;
        SRL_HIP_CHECK(h, hipStreamSynchronize(h->stream));
        return 0;
    }
    hipLaunchKernelGGL(kuka_refresh_k, dim3((h->n + 63) / 64), dim3(64), 0, h->stream, *h->kuka, h->n);
    SRL_HIP_CHECK(h, hipGetLastError());
    SRL_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    return 0;
}

int kuka_field(Handle *h, int field, void **dptr, size_t *elem, int *count) {
    KukaState *s = h->kuka;
    const size_t n = (size_t)h->n;
    *elem = 8; *count = 1;
    switch (field) {
        case SRLHIP_F_KUKA_Q: *dptr = s->d + D_Q * n; *count = 7; return 0;
        case SRLHIP_F_KUKA_QD: *dptr = s->d + D_QD * n; *count = 7; return 0;
        case SRLHIP_F_KUKA_EE_TARGET: *dptr = s->d + D_EE * n; *count = 3; return 0;
        case SRLHIP_F_KUKA_BUTTON_Q: *dptr = s->d + D_BQ * n; *count = 2; return 0;
        case SRLHIP_F_KUKA_BUTTON_POS: *dptr = s->d + D_BPOS * n; *count = 3; return 0;
        case SRLHIP_F_KUKA_GRIPPER: *dptr = s->d + D_GRIP * n; *count = 3; return 0;
        case SRLHIP_F_STEP_COUNT: *dptr = s->i + I_COUNTER * n; *elem = 4; return 0;
        case SRLHIP_F_KUKA_COUNTERS: *dptr = s->i + I_NCONTACT * n; *elem = 4; *count = 3; return 0;
        case SRLHIP_F_KUKA_BUTTON_XY: *dptr = s->d + D_BX * n; *count = 2; return 0;
        case SRLHIP_F_KUKA_BUTTON2_Q: *dptr = s->d + D_B2Q * n; *count = 2; return 0;
        case SRLHIP_F_KUKA_BUTTON2_XY: *dptr = s->d + D_B2X * n; *count = 2; return 0;
        case SRLHIP_F_KUKA_GOAL: *dptr = s->i + I_GOAL * n; *elem = 4; *count = 2; return 0;
        case SRLHIP_F_KUKA_OBJECTS: *dptr = s->objs; *count = 30; return 0;
        case SRLHIP_F_KUKA_BODIES:
            if (!s->full || h->cfg.env_kind != SRLHIP_ENV_KUKA_RAND) return h->fail(SRLHIP_EINVAL, "KUKA_BODIES: free bodies exist on full-model KukaRandButtonGymEnv handles only");
            *dptr = s->rb; *count = 66; return 0;
        case SRLHIP_F_KUKA_IK_CROSSED:
            if (!s->full) return h->fail(SRLHIP_EINVAL, "KUKA_IK_CROSSED: the IK conditioning flag exists on full-model handles only (cfg.kuka_model = SRLHIP_KUKA_MODEL_FULL)");
            *dptr = s->i + I_IKX * n; *elem = 4; return 0;
        case SRLHIP_F_KUKA_GRIPPER_Q:
        case SRLHIP_F_KUKA_GRIPPER_QD:
            // the lumped model has no gripper DoFs: its kernels never write these planes
            if (!s->full) return h->fail(SRLHIP_EINVAL, "KUKA_GRIPPER_Q / KUKA_GRIPPER_QD exist on full-model handles only (cfg.kuka_model = SRLHIP_KUKA_MODEL_FULL)");
            *dptr = s->d + (field == SRLHIP_F_KUKA_GRIPPER_Q ? D_GQ : D_GQD) * n; *count = 5; return 0;
    }
    return h->fail(SRLHIP_EINVAL, "unknown field for KukaButtonGymEnv");
}

}  // namespace srl
