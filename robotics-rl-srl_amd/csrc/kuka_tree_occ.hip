// kuka_tree_occ.hip — the full-model lane-group rollout at TWO wavefronts per SIMD, for very large batches (one-button envs, Cartesian
// action modes; the library picks it from 65536 envs up, SRLHIP_KUKA_OCC=0|1 forces either variant).  What it takes (kuka_tree.hpp,
// OCC = 1): <= 256 registers (what the one-wavefront kernel keeps in its other 256 becomes 0.8-1.0 KB per lane of scratch) and <= 20 KiB
// of LDS per wavefront — ONE general-path work area per wavefront, taken by its four envs in turns, a 2.1 KiB park per env, sphere /
// limit candidates recomputed inside the turn; two wavefronts share a workgroup and its lane table.
// What it buys is bounded: the projected Gauss-Seidel sweep is one dependent chain of float64 ops, and one wavefront alone already issues
// such a chain every 6.2 cycles where the SIMD's limit is 4.1 (profiles/probes/f64_issue_rate.hip, profiles/r04_f64_issue_rate.txt): two
// chains on one SIMD reach 1.31x at best.  Measured whole-rollout gain (profiles/r04_occ_nsweep_final.jsonl, after the contact-sweep work):
// -15 % at 16384 envs, -5 % at 32768, +1 % at 65536, +4 % at 131072 (before that work: +2 % / +6 % / +9 % from 32768, r04_occ_nsweep.jsonl).
#include "kuka_tree_kernels.hpp"

namespace srl {
using namespace kuka;
namespace {
constexpr int kOccBlock = 128;

// T consecutive VecEnv steps per launch, two wavefronts per SIMD.  GIVEN: the caller supplies the actions (a compile-time switch: a possible action load
// inside the step loop makes every step wait for the previous step's output stores — gfx9 counts loads and stores together).
template <int MODE, bool GIVEN>
__global__ void __attribute__((amdgpu_flat_work_group_size(kOccBlock, kOccBlock), amdgpu_waves_per_eu(2, 2)))
kuka_tree_rollout_occ_k(KukaParams p, KukaState s, RngState rs, EpisodeStats st, int T, const void *actions, const double *noise,
                    float *obs, float *rew, uint8_t *done_out, void *act_out) {
    using namespace grp;
    constexpr int NB = 1, RB = 0;
    __shared__ double work_all[kOccBlock / 64][tree::kTreeWorkDoubles];        // one general-path work area per wavefront
    __shared__ double park_all[kOccBlock / GL][tree::kTreeParkDoubles];        // per env: the own row of M^-1, the spatial axes
    const int64_t n = p.n;
    const int e_raw = blockIdx.x * (kOccBlock / GL) + (int)(threadIdx.x / GL);
    const bool valid = e_raw < p.n;
    const int e = valid ? e_raw : p.n - 1;           // tail groups shadow the last env (every lane stays active for the cross-lane ops)
    const Cfg &cfg = p.cfg;
    __shared__ double tab[tree::kLaneTableDoubles];
    double *scratch = work_all[threadIdx.x / 64], *park = park_all[threadIdx.x / GL];
    LaneId L;
    build_lane_table(L, s.ttable, tab);
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
    tree::tfk(tree::lane_view(tab), g);
    // (Monitor's record of the last finished episode is written only when an episode finishes in this launch and is not read: on
    //  host-pointer handles those two planes are mapped host memory — srlhip_episode_records — and a read / an unconditional
    //  write-back would cross PCIe in every launch)
    double ep_ret = st.ep_return[e], last_ret = 0.0, last_reward = 0.0;
    int32_t ep_len = st.ep_length[e], last_len = 0, n_fin = st.n_finished[e];
    const int32_t n_fin0 = n_fin;
    GroupActions gact; gact.init(rs.key[e], rs.key[n + e], rs.act_ctr[e]);
    Philox &act = gact.p;
    const int od = cfg.obs_mode == 1 ? 14 : cfg.obs_mode == 2 ? 17 : 3;
    const int adim = cfg.is_discrete ? 1 : cfg.action_joints ? 7 : 3;
    // Every load of the prologue retires HERE.  Otherwise the compiler's wait for them sits at their first use INSIDE the loop, as
    // `s_waitcnt vmcnt(0)` — and gfx9 counts stores on the same counter, so from the second step on that wait drains the previous
    // step's output stores: a full store round trip (~1.5 k cycles) in every step of the rollout.
    __builtin_amdgcn_s_waitcnt(0x0F70);
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
        reward = tree::tenv_step<NB, RB, 1>(v, g, tab, cfg, scratch, rng0, a, ca, ca_own, &done, &body, park);
        ep_ret += reward; ep_len += 1; last_reward = reward;
        const int info = cfg.info_bits ? (v.ikx & 1) << 1 : 0;      // (kuka_tree_kernels.hpp)
        if (done) {
            last_ret = ep_ret; last_len = ep_len; n_fin += 1; ep_ret = 0.0; ep_len = 0;
            if (cfg.auto_reset) {
                double *objs = valid ? s.objs + e : nullptr;
                tree::tenv_reset<0, NB, RB, 1>(v, g, tab, cfg, scratch, rng0, s.tstarts, s.tsettled, objs, n, &body, park);
                // the start-state loads retire HERE, not at their first use in the next step (where vmcnt(0) would also wait for
                // the output stores of steps that did not reset)
                __builtin_amdgcn_s_waitcnt(0x0F70);
            }
        }
        if (lead) {
            if (obs) observe(v, cfg, obs + row * od, 1);
            if (rew) rew[row] = (float)reward;
            if (done_out) done_out[row] = (uint8_t)((int)done | info);
        }
    }
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

}  // namespace

#define SRL_TREE_OCC_GO(MODE, G) hipLaunchKernelGGL((kuka_tree_rollout_occ_k<MODE, G>), grid, block, 0, h->stream, p, *h->kuka, h->rng, h->stats, T, d_actions, d_noise, obs, d_rew, d_done, d_act_out)
int kuka_tree_occ_launch(Handle *h, const KukaParams &p, int T, const void *d_actions, const double *d_noise, float *obs, float *d_rew,
                         uint8_t *d_done, void *d_act_out) {
    const int envs_per_block = kOccBlock / grp::GL;
    dim3 grid((h->n + envs_per_block - 1) / envs_per_block), block(kOccBlock);
    switch (h->cfg.rng_mode) {
        case SRLHIP_RNG_PHILOX: if (d_actions) SRL_TREE_OCC_GO(SRLHIP_RNG_PHILOX, true); else SRL_TREE_OCC_GO(SRLHIP_RNG_PHILOX, false); break;
        case SRLHIP_RNG_MT19937: if (d_actions) SRL_TREE_OCC_GO(SRLHIP_RNG_MT19937, true); else SRL_TREE_OCC_GO(SRLHIP_RNG_MT19937, false); break;
        default: if (d_actions) SRL_TREE_OCC_GO(SRLHIP_RNG_HOST, true); else SRL_TREE_OCC_GO(SRLHIP_RNG_HOST, false);
    }
    SRL_HIP_CHECK(h, hipGetLastError());
    return 0;
}

}  // namespace srl
