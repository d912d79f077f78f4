// kuka_group_kernels.hpp — the lane-group kernels (templates), included by kuka_group.hip (CM = false instantiations) and
// kuka_group_cm.hip (CM = true instantiations + reset / settle / self-test kernels).
#pragma once
#include "kuka_device.hpp"
#include "kuka_group.hpp"

namespace srl {
namespace {

// ---------------------------------------------------------------------------------------------------------------------
// Lane-group rollout (kuka_group.hpp): 16 lanes = one DPP row per env, one wavefront (4 envs) per workgroup.  A 4096-env
// batch is 1024 wavefronts — one per SIMD of the MI355X — instead of the 64 of kuka_rollout_k; same state planes, same
// start-state table, same outputs.  KukaButton / MovingButton / RandButton (NB == 1).

// GIVEN: the caller supplies the actions.  A compile-time switch because a possible action load inside the step loop makes
// the compiler wait for vmcnt(0) every step — which on gfx9 also waits for the previous step's output STORES to retire
// (loads and stores share the counter): the random-agent variant has no load in its loop and never waits on memory.
// CM: a runtime model table is installed (srlhip_set_kuka_model): per-lane constants come from s.model, resets integrate
// their init actions from the settled state instead of reading the start table.
template <int MODE, bool JOINTS, bool GIVEN, bool CM>
__global__ void __launch_bounds__(kGroupBlock)
kuka_group_rollout_k(KukaParams p, KukaState s, RngState rs, EpisodeStats st, int T, const void *actions, const double *noise,
                     float *obs, float *rew, uint8_t *done_out, void *act_out) {
    using namespace grp;
    __shared__ double scratch_all[kGroupEnvs][kScratchDoubles];
    const int64_t n = p.n;
    const int e_raw = blockIdx.x * kGroupEnvs + (int)(threadIdx.x / GL);
    const bool valid = e_raw < p.n;
    const int e = valid ? e_raw : p.n - 1;           // tail groups shadow the last env (every lane stays active for the DPP ops)
    const Cfg &cfg = p.cfg;
    double *scratch = scratch_all[threadIdx.x / GL];
    Lane L;
    lane_init<CM>(L, s.model);
    const bool lead = L.l == 0 && valid;
    // Philox mode: the lane-group stream adaptor (batched Gaussian draws); otherwise the generators of the lane-per-env kernel
    using Rng = std::conditional_t<MODE == SRLHIP_RNG_PHILOX, GroupPhilox, typename KRng<MODE>::type>;
    Rng rng0;
    if constexpr (MODE == SRLHIP_RNG_PHILOX) rng0.init(rs.key[e], rs.key[n + e], rs.ctr[e]);
    else krng_load<MODE>(rng0, rs, e, p.n, noise ? noise + e : nullptr);
    // generators whose state lives in HBM (MT19937) are advanced by lane 0 only; counter-based / host streams are replayed by all
    Lane0Rng<Rng> rng_l0{&rng0, lead};
    Env v = {};
    GState g;
    {   // env scalars replicated on the row, the own joint per arm lane
#pragma unroll
        for (int k = 0; k < 3; k++) { v.ee[k] = s.d[(D_EE + k) * n + e]; v.bpos[k] = s.d[(D_BPOS + k) * n + e]; v.grip[k] = s.d[(D_GRIP + k) * n + e]; }
        v.bq = s.d[D_BQ * n + e]; v.bqd = s.d[D_BQD * n + e]; v.bx = s.d[D_BX * n + e]; v.by = s.d[D_BY * n + e];
        v.bz = s.d[D_BZ * n + e]; v.bspeed = s.d[D_BSPEED * n + e];
        v.motor_on = s.i[I_MOTOR * n + e]; v.contact_button = s.i[I_CB * n + e]; v.contact_table = s.i[I_CT * n + e];
        v.counter = s.i[I_COUNTER * n + e]; v.n_contacts = s.i[I_NCONTACT * n + e]; v.n_outside = s.i[I_NOUT * n + e];
        v.terminated = s.i[I_TERM * n + e];
        const int j = L.arm ? L.l : 0;
        g.q = s.d[(D_Q + j) * n + e]; g.qd = s.d[(D_QD + j) * n + e]; g.sq = s.d[(D_SQ + j) * n + e]; g.cq = s.d[(D_CQ + j) * n + e];
        if (!L.arm) { g.q = 0.0; g.qd = 0.0; g.sq = 0.0; g.cq = 1.0; }
        gfk<CM>(L, g);
    }
    double ep_ret = st.ep_return[e], last_ret = 0.0, last_reward = 0.0;      // (last_* : written only when an episode finished, never read — mobile.hip)
    int32_t ep_len = st.ep_length[e], last_len = 0, n_fin = st.n_finished[e];
    const int32_t n_fin0 = n_fin;
    GroupActions gact; gact.init(rs.key[e], rs.key[n + e], rs.act_ctr[e]);
    Philox &act = gact.p;
    const int od = cfg.obs_mode == 1 ? 14 : cfg.obs_mode == 2 ? 17 : 3;
    const int adim = cfg.is_discrete ? 1 : cfg.action_joints ? 7 : 3;
    for (int t = 0; t < T; t++) {
        const int64_t row = (int64_t)t * n + e;
        int a = 0; float ca[7] = {0, 0, 0, 0, 0, 0, 0};
        if constexpr (GIVEN) {
            if (cfg.is_discrete) a = static_cast<const int32_t *>(actions)[row];
            else for (int j = 0; j < adim; j++) ca[j] = static_cast<const float *>(actions)[row * adim + j];
        } else {
            if (cfg.is_discrete) a = gact.next(5);
            else for (int j = 0; j < adim; j += 2) {
                uint32_t o[4]; act.block(o);
                ca[j] = (float)(-1.0 + 2.0 * Philox::to_double(o[0], o[1]));
                if (j + 1 < adim) ca[j + 1] = (float)(-1.0 + 2.0 * Philox::to_double(o[2], o[3]));
            }
            if (act_out && lead) {
                if (cfg.is_discrete) static_cast<int32_t *>(act_out)[row] = a;
                else for (int j = 0; j < adim; j++) static_cast<float *>(act_out)[row * adim + j] = ca[j];
            }
        }
        float ca_own = 0.f;
#pragma unroll
        for (int j = 0; j < ND; j++) ca_own = L.l == j ? ca[j] : ca_own;
        bool done;
        double reward;
        if constexpr (MODE == SRLHIP_RNG_MT19937) reward = genv_step<CM>(v, g, L, cfg, scratch, rng_l0, a, ca, ca_own, &done);
        else reward = genv_step<CM>(v, g, L, cfg, scratch, rng0, a, ca, ca_own, &done);
        ep_ret += reward; ep_len += 1; last_reward = reward;
        if (done) {
            last_ret = ep_ret; last_len = ep_len; n_fin += 1; ep_ret = 0.0; ep_len = 0;
            if (cfg.auto_reset) {
                double *objs = valid ? s.objs + e : nullptr;
                if constexpr (MODE == SRLHIP_RNG_MT19937) genv_reset<JOINTS, CM>(v, g, L, cfg, scratch, rng_l0, s.starts, s.settled, objs, n);
                else genv_reset<JOINTS, CM>(v, g, L, cfg, scratch, rng0, s.starts, s.settled, objs, n);
                // the start-state loads retire HERE: otherwise the wait for them lands at their first use in the next step, on
                // every path, where vmcnt(0) also waits for the output stores of steps that did not reset
                __builtin_amdgcn_s_waitcnt(0x0F70);
            }
        }
        if (lead) {
            if (obs) observe(v, cfg, obs + row * od, 1);
            if (rew) rew[row] = (float)reward;
            if (done_out) done_out[row] = (uint8_t)done;
        }
    }
    // (the exit stores recompute their plane addresses from an opaque copy of the env index: otherwise the ~25 addresses
    //  formed for the entry loads stay live across the whole rollout loop and push its working set into scratch)
    int e_out = e;
    asm volatile("" : "+v"(e_out));
    const int e_in = e;
    (void)e_in;
#define e e_out
    if (valid && L.arm) {
        s.d[(D_Q + L.l) * n + e] = g.q; s.d[(D_QD + L.l) * n + e] = g.qd; s.d[(D_SQ + L.l) * n + e] = g.sq; s.d[(D_CQ + L.l) * n + e] = g.cq;
    }
    if (lead) {
#pragma unroll
        for (int k = 0; k < 3; k++) { s.d[(D_EE + k) * n + e] = v.ee[k]; s.d[(D_BPOS + k) * n + e] = v.bpos[k]; s.d[(D_GRIP + k) * n + e] = v.grip[k]; }
        s.d[D_BQ * n + e] = v.bq; s.d[D_BQD * n + e] = v.bqd; s.d[D_BX * n + e] = v.bx; s.d[D_BY * n + e] = v.by;
        s.d[D_BZ * n + e] = v.bz; s.d[D_BSPEED * n + e] = v.bspeed;
        s.i[I_MOTOR * n + e] = v.motor_on; s.i[I_CB * n + e] = v.contact_button; s.i[I_CT * n + e] = v.contact_table;
        s.i[I_COUNTER * n + e] = v.counter; s.i[I_NCONTACT * n + e] = v.n_contacts; s.i[I_NOUT * n + e] = v.n_outside;
        s.i[I_TERM * n + e] = v.terminated;
        if constexpr (MODE == SRLHIP_RNG_PHILOX) rs.ctr[e] = rng0.p.ctr;
        else krng_store<MODE>(rng0, rs, e);
        if constexpr (!GIVEN) rs.act_ctr[e] = act.ctr;
        st.ep_return[e] = ep_ret; st.ep_length[e] = ep_len;
        if (n_fin != n_fin0) { st.last_return[e] = last_ret; st.last_length[e] = last_len; }
        st.n_finished[e] = n_fin; st.last_reward[e] = last_reward;
    }
#undef e
}


}  // namespace

// one launcher body for both translation units: CM is the unit's compile-time model switch
#define SRL_GROUP_LAUNCHER(NAME, CM)                                                                                                 \
    int NAME(Handle *h, const KukaParams &p, int T, const void *d_actions, const double *d_noise, float *obs, float *d_rew,         \
             uint8_t *d_done, void *d_act_out) {                                                                                     \
        dim3 grid((h->n + kGroupEnvs - 1) / kGroupEnvs), block(kGroupBlock);                                                         \
        const bool joints = !h->cfg.is_discrete && h->cfg.action_joints;                                                             \
        switch (h->cfg.rng_mode) {                                                                                                   \
            case SRLHIP_RNG_PHILOX: SRL_GROUP_MODE(SRLHIP_RNG_PHILOX, CM) break;                                                     \
            case SRLHIP_RNG_MT19937: SRL_GROUP_MODE(SRLHIP_RNG_MT19937, CM) break;                                                   \
            default: SRL_GROUP_MODE(SRLHIP_RNG_HOST, CM)                                                                             \
        }                                                                                                                            \
        SRL_HIP_CHECK(h, hipGetLastError());                                                                                         \
        return 0;                                                                                                                    \
    }
#define SRL_GROUP_GO(MODE, J, G, CM) hipLaunchKernelGGL((kuka_group_rollout_k<MODE, J, G, CM>), grid, block, 0, h->stream, p, *h->kuka, h->rng, h->stats, T, d_actions, d_noise, obs, d_rew, d_done, d_act_out)
#define SRL_GROUP_MODE(MODE, CM)                                                            \
    if (joints && d_actions) SRL_GROUP_GO(MODE, true, true, CM);                            \
    else if (joints) SRL_GROUP_GO(MODE, true, false, CM);                                   \
    else if (d_actions) SRL_GROUP_GO(MODE, false, true, CM);                                \
    else SRL_GROUP_GO(MODE, false, false, CM);

}  // namespace srl
