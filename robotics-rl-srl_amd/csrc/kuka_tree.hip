// kuka_tree.hip — lane-group kernels of the FULL Kuka model (kuka_tree.hpp: 12-DoF arm + gripper tree, per-link contact spheres,
// friction rows): rollout / step, reset, settle + start-state table, refresh after srlhip_set_state.  Launch geometry of the
// lane-group family: 16 lanes = one DPP row per env, one wavefront (4 envs) per workgroup — a 4096-env batch is 1024 wavefronts,
// one per SIMD of the MI355X.  The model is a table in HBM read once per launch into per-lane registers (srlhip_kuka_tree_model).
#include "kuka_tree_kernels.hpp"

namespace srl {
using namespace kuka;

namespace {
__global__ void __launch_bounds__(kGroupBlock) kuka_tree_table_k(const TreeModel *m, double *ttable) {
    TLane L;
    tree::lane_init(L, m);
    if (threadIdx.x < grp::GL) tree::lane_store(L, ttable);
}

// 500 settle steps (kuka_button_gym_env.py:242-247): every group of the wavefront integrates the same env, group 0 publishes
__global__ void __launch_bounds__(kGroupBlock) kuka_tree_settle_k(KukaParams p, KukaState s) {
    using namespace grp;
    __shared__ double scratch_all[kGroupEnvs][kTS];
    __shared__ double tab[tree::kLaneTableDoubles];
    LaneId L; build_lane_table(L, s.ttable, tab);
    Env e = {};
    GState g;
    tree::tinitial(e, g, tab);
    const double zero[3] = {0, 0, 0};
    for (int i = 0; i < kNSettleSteps; i++) tree::tphysics_step(e, g, tab, p.cfg, scratch_all[threadIdx.x / GL], zero, p.cfg.action_joints != 0, L.q0, 0.0);
    if (threadIdx.x < GL) tree::tpack_start(e, g, s.tsettled);
}
// table of the 6^5 (2^5) possible episode start states: one lane group per state
__global__ void __launch_bounds__(kGroupBlock) kuka_tree_starts_k(KukaParams p, KukaState s) {
    using namespace grp;
    __shared__ double scratch_all[kGroupEnvs][kTS];
    const int idx_raw = blockIdx.x * kGroupEnvs + (int)(threadIdx.x / GL);
    const bool valid = idx_raw < s.nstarts;
    const int idx = valid ? idx_raw : s.nstarts - 1;
    __shared__ double tab[tree::kLaneTableDoubles];
    LaneId L; build_lane_table(L, s.ttable, tab);
    Env e = {};
    GState g;
    tree::tunpack_start(e, g, s.tsettled);
    e.bx = kButtonX; e.by = kButtonY; e.bz = L.base_z;
    tree::tfk(tree::lane_view(tab), g);
    const int base = p.cfg.is_discrete ? 6 : 2;
    int rem = idx;
    double motor[3];
    for (int k = 0; k < kNInitActions; k++) {
        init_action_motor(p.cfg, rem % base, motor);
        tree::tphysics_step(e, g, tab, p.cfg, scratch_all[threadIdx.x / GL], motor, false, L.q0, 0.0);
        rem /= base;
    }
    if (valid) tree::tpack_start(e, g, s.tstarts + (int64_t)idx * kTStart);
}
// after srlhip_set_state(KUKA_Q / GRIPPER_Q): refresh the cached sin / cos and the gripper position
__global__ void __launch_bounds__(kGroupBlock) kuka_tree_refresh_k(KukaParams p, KukaState s) {
    using namespace grp;
    const int64_t n = p.n;
    const int e_raw = blockIdx.x * kGroupEnvs + (int)(threadIdx.x / GL);
    const bool valid = e_raw < p.n;
    const int e = valid ? e_raw : p.n - 1;
    __shared__ double tab[tree::kLaneTableDoubles];
    LaneId L; build_lane_table(L, s.ttable, tab);
    Env v = {};
    GState g;
    tload(s, n, e, L, v, g, p.cfg.two != 0);
    tree::trefresh(tree::lane_view(tab), g, v);
    tstore(s, n, e, L, v, g, valid, p.cfg.two != 0);
}

}  // namespace

#define SRL_TREE_GO(MODE, J, G, NB) hipLaunchKernelGGL((kuka_tree_rollout_k<MODE, J, G, NB>), grid, block, 0, h->stream, p, *h->kuka, h->rng, h->stats, T, d_actions, d_noise, obs, d_rew, d_done, d_act_out, G ? sig : PersistArgs{})
// (Kuka2ButtonGymEnv takes discrete actions only: no joints-mode instantiation of the two-button kernels)
#define SRL_TREE_MODE(MODE)                                         \
    if (two && d_actions) SRL_TREE_GO(MODE, false, true, 2);        \
    else if (two) SRL_TREE_GO(MODE, false, false, 2);               \
    else if (joints && d_actions) SRL_TREE_GO(MODE, true, true, 1); \
    else if (joints) SRL_TREE_GO(MODE, true, false, 1);             \
    else if (d_actions) SRL_TREE_GO(MODE, false, true, 1);          \
    else SRL_TREE_GO(MODE, false, false, 1);
int kuka_tree_launch(Handle *h, const KukaParams &p, int T, const void *d_actions, const double *d_noise, float *obs, float *d_rew,
                     uint8_t *d_done, void *d_act_out) {
    if (h->cfg.env_kind == SRLHIP_ENV_KUKA_RAND) return kuka_tree_rb_launch(h, p, T, d_actions, d_noise, obs, d_rew, d_done, d_act_out);   // free bodies: kuka_tree_rb.hip
    // the reference's default KukaButtonGymEnv configuration on a device RNG mode: the instantiation with that configuration folded in
    // (kuka_tree_kernels.hpp, SPEC = 1); SRLHIP_KUKA_SPEC=0 keeps the generic instantiation (tests compare the two)
    const srlhip_config &c = h->cfg;
    static const bool spec_enabled = [] { const char *v = getenv("SRLHIP_KUKA_SPEC"); return !v || atoi(v) != 0; }();
    const bool spec = spec_enabled && reinterpret_cast<const TreeModel *>(h->kuka_tmodel_host)->solver_detail == 0.0 && c.env_kind == SRLHIP_ENV_KUKA_BUTTON && (c.rng_mode == SRLHIP_RNG_PHILOX || c.rng_mode == SRLHIP_RNG_MT19937) && c.is_discrete && !c.action_joints && !c.random_target &&
                      c.force_down && !c.shape_reward && c.action_repeat == 1 && c.auto_reset &&
                      (c.obs_mode == SRLHIP_OBS_GROUND_TRUTH || (c.obs_mode == SRLHIP_OBS_RAW_PIXELS && !obs));      // raw_pixels: the rasteriser draws, the stepper writes no observation
    {
        // Very large batches run the two-wavefronts-per-SIMD variant (kuka_tree_occ.hip: one-button envs, Cartesian actions).
        // Measured after the contact-sweep work (profiles/r04_occ_nsweep_final.jsonl, one -> two wavefronts, env-steps/s x 1e8):
        // 16384 envs 1.62 -> 1.37, 32768 1.63 -> 1.56, 65536 1.65 -> 1.67, 131072 1.66 -> 1.73 (before that work the variant won from
        // 32768: profiles/r04_occ_nsweep.jsonl).  The sweep is a dependent f64 chain that one wavefront already issues at 6.2 of
        // the SIMD's 4.1 cycles per op (profiles/r04_f64_issue_rate.txt), so a second wavefront can hide little, its register diet
        // (256 instead of 512) costs ~1 KB/lane of scratch, and its four envs take turns in the contact path.  Hence the threshold.
        // SRLHIP_KUKA_OCC=0|1 forces either variant (tests run both on small batches).
        const char *force = getenv("SRLHIP_KUKA_OCC");
        const bool one_button = h->cfg.env_kind == SRLHIP_ENV_KUKA_BUTTON || h->cfg.env_kind == SRLHIP_ENV_KUKA_MOVING;
        const bool cartesian = h->cfg.is_discrete || !h->cfg.action_joints;
        // (round 5: where the configuration-specialised one-wavefront instantiation applies it is the faster one at every size measured
        //  — 65536 envs 1.73e8 against 1.66e8, 131072 1.74e8 against 1.72e8, profiles/r05_nsweep_kuka.jsonl — so the default dispatch takes
        //  the two-wavefront variant only for configurations that instantiation does not cover)
        const bool occ = force ? force[0] == '1' : (h->n >= 65536 && !spec);
        if (occ && one_button && cartesian) return kuka_tree_occ_launch(h, p, T, d_actions, d_noise, obs, d_rew, d_done, d_act_out);
    }
    dim3 grid(((h->n + kGroupEnvs - 1) / kGroupEnvs + 7) / 8 * 8), block(kGroupBlock);      // a multiple of 8: the rollout kernel maps blocks to envs XCD by XCD
    const bool joints = !h->cfg.is_discrete && h->cfg.action_joints, two = h->cfg.env_kind == SRLHIP_ENV_KUKA_2BUTTON;
    // a single-step launch with the caller's actions on a handle whose host side armed the early completion signal (api.hip): this
    // kernel reports the step's outputs per eighth of the grid before it ends
    PersistArgs sig{};
    if (h->step_signal && T == 1 && d_actions) {
        sig = *h->step_signal; h->step_signal_armed = true;
        const uint32_t blocks = (uint32_t)(h->n + kGroupEnvs - 1) / kGroupEnvs, per = (blocks + 7) / 8;      // eighth g = the contiguous workgroup range [g, g + 1) * per
        h->signal_eighths = 0;
        for (uint32_t g = 0; g < 8 && g * per < blocks; g++) h->signal_eighths |= 1u << g;
    }
    if (spec) {
#define SRL_TREE_SPEC(MODE, G) hipLaunchKernelGGL((kuka_tree_rollout_k<MODE, false, G, 1, 0, 1>), grid, block, 0, h->stream, p, *h->kuka, h->rng, h->stats, T, d_actions, d_noise, obs, d_rew, d_done, d_act_out, G ? sig : PersistArgs{})
        if (c.rng_mode == SRLHIP_RNG_PHILOX) { if (d_actions) SRL_TREE_SPEC(SRLHIP_RNG_PHILOX, true); else SRL_TREE_SPEC(SRLHIP_RNG_PHILOX, false); }
        else { if (d_actions) SRL_TREE_SPEC(SRLHIP_RNG_MT19937, true); else SRL_TREE_SPEC(SRLHIP_RNG_MT19937, false); }     // (the reference's own streams: HipVecEnv's default)
#undef SRL_TREE_SPEC
        SRL_HIP_CHECK(h, hipGetLastError());
        return 0;
    }
    switch (h->cfg.rng_mode) {
        case SRLHIP_RNG_PHILOX: SRL_TREE_MODE(SRLHIP_RNG_PHILOX) break;
        case SRLHIP_RNG_MT19937: SRL_TREE_MODE(SRLHIP_RNG_MT19937) break;
        default: SRL_TREE_MODE(SRLHIP_RNG_HOST)
    }
    SRL_HIP_CHECK(h, hipGetLastError());
    return 0;
}

// ---- persistent stepping (internal.hpp PersistArgs; srlhip_set_persistent) ---------------------------------------------------------
// Persistent instantiations exist for KukaButtonGymEnv, KukaMovingButtonGymEnv and Kuka2ButtonGymEnv on a device RNG mode: the
// configuration-specialised kernel where kuka_tree_launch would pick it (reference ctor defaults, default solver details), the generic
// one otherwise (random_target, shaped reward, continuous Cartesian or joint-space actions, the joints observation modes, other
// solver details).  The grid must be CO-RESIDENT: a workgroup that waits for the host in a loop never makes room for one that has not
// started.
static int persist_kind(const Handle *h) {          // 0: none, 1: SPEC, 2: generic Cartesian, 3: generic joint-space actions, 4: Kuka2Button
    const srlhip_config &c = h->cfg;
    if (!h->kuka || c.kuka_model != SRLHIP_KUKA_MODEL_FULL || c.env_kind == SRLHIP_ENV_KUKA_RAND ||      // (KukaRandButton: kuka_tree_rb.hip, kinds 5, 6)
        !(c.rng_mode == SRLHIP_RNG_PHILOX || c.rng_mode == SRLHIP_RNG_MT19937) || c.obs_mode == SRLHIP_OBS_RAW_PIXELS)
        return 0;
    if (c.env_kind == SRLHIP_ENV_KUKA_2BUTTON) return c.is_discrete ? 4 : 0;          // (Kuka2ButtonGymEnv takes discrete actions only)
    const bool spec = reinterpret_cast<const TreeModel *>(h->kuka_tmodel_host)->solver_detail == 0.0 && c.env_kind == SRLHIP_ENV_KUKA_BUTTON && c.is_discrete &&
                      !c.action_joints && !c.random_target && c.force_down && !c.shape_reward && c.action_repeat == 1 && c.auto_reset && c.obs_mode == SRLHIP_OBS_GROUND_TRUTH;
    if (spec) return 1;
    return (!c.is_discrete && c.action_joints) ? 3 : 2;
}
static const void *persist_fn(const Handle *h, int kind) {
    const bool ph = h->cfg.rng_mode == SRLHIP_RNG_PHILOX;
    switch (kind) {
        case 1: return ph ? reinterpret_cast<const void *>(kuka_tree_rollout_k<SRLHIP_RNG_PHILOX, false, true, 1, 0, 1, 1>) : reinterpret_cast<const void *>(kuka_tree_rollout_k<SRLHIP_RNG_MT19937, false, true, 1, 0, 1, 1>);
        case 2: return ph ? reinterpret_cast<const void *>(kuka_tree_rollout_k<SRLHIP_RNG_PHILOX, false, true, 1, 0, 0, 1>) : reinterpret_cast<const void *>(kuka_tree_rollout_k<SRLHIP_RNG_MT19937, false, true, 1, 0, 0, 1>);
        case 3: return ph ? reinterpret_cast<const void *>(kuka_tree_rollout_k<SRLHIP_RNG_PHILOX, true, true, 1, 0, 0, 1>) : reinterpret_cast<const void *>(kuka_tree_rollout_k<SRLHIP_RNG_MT19937, true, true, 1, 0, 0, 1>);
        case 4: return ph ? reinterpret_cast<const void *>(kuka_tree_rollout_k<SRLHIP_RNG_PHILOX, false, true, 2, 0, 0, 1>) : reinterpret_cast<const void *>(kuka_tree_rollout_k<SRLHIP_RNG_MT19937, false, true, 2, 0, 0, 1>);
    }
    return nullptr;
}
int kuka_tree_persist_blocks(Handle *h, int *capacity) {
    if (capacity) *capacity = 0;
    const int kind = persist_kind(h);
    if (!kind) return 0;
    const int real = (h->n + kGroupEnvs - 1) / kGroupEnvs, grid = (real + 7) / 8 * 8;
    int per_cu = 0;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, h->cfg.device_id) != hipSuccess) return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, persist_fn(h, kind), kGroupBlock, 0) != hipSuccess) return 0;
    if (capacity) *capacity = per_cu * prop.multiProcessorCount;       // workgroups of this kernel the device holds at once
    return (long long)per_cu * prop.multiProcessorCount >= grid ? real : 0;
}
int kuka_tree_persist_launch(Handle *h, const KukaParams &p, const void *d_actions, float *obs, float *d_rew, uint8_t *d_done, const PersistArgs &pa) {
    dim3 grid(((h->n + kGroupEnvs - 1) / kGroupEnvs + 7) / 8 * 8), block(kGroupBlock);
    const double *no_noise = nullptr;
    void *no_act = nullptr;
    const bool ph = h->cfg.rng_mode == SRLHIP_RNG_PHILOX;
#define SRL_TREE_PERSIST(MODE, J, SPEC) hipLaunchKernelGGL((kuka_tree_rollout_k<MODE, J, true, 1, 0, SPEC, 1>), grid, block, 0, h->stream, p, *h->kuka, h->rng, h->stats, 0, d_actions, no_noise, obs, d_rew, d_done, no_act, pa)
#define SRL_TREE_PERSIST2(MODE) hipLaunchKernelGGL((kuka_tree_rollout_k<MODE, false, true, 2, 0, 0, 1>), grid, block, 0, h->stream, p, *h->kuka, h->rng, h->stats, 0, d_actions, no_noise, obs, d_rew, d_done, no_act, pa)
    switch (persist_kind(h)) {
        case 1: if (ph) SRL_TREE_PERSIST(SRLHIP_RNG_PHILOX, false, 1); else SRL_TREE_PERSIST(SRLHIP_RNG_MT19937, false, 1); break;
        case 2: if (ph) SRL_TREE_PERSIST(SRLHIP_RNG_PHILOX, false, 0); else SRL_TREE_PERSIST(SRLHIP_RNG_MT19937, false, 0); break;
        case 3: if (ph) SRL_TREE_PERSIST(SRLHIP_RNG_PHILOX, true, 0); else SRL_TREE_PERSIST(SRLHIP_RNG_MT19937, true, 0); break;
        case 4: if (ph) SRL_TREE_PERSIST2(SRLHIP_RNG_PHILOX); else SRL_TREE_PERSIST2(SRLHIP_RNG_MT19937); break;
        default: return h->fail(SRLHIP_ENOTSUP, "persistent stepping: no instantiation for this configuration");
    }
#undef SRL_TREE_PERSIST
#undef SRL_TREE_PERSIST2
    SRL_HIP_CHECK(h, hipGetLastError());
    return 0;
}

int kuka_tree_reset(Handle *h, const KukaParams &p, const uint8_t *d_mask, const double *d_host_rand, int stride, float *obs) {
    if (h->cfg.env_kind == SRLHIP_ENV_KUKA_RAND) return kuka_tree_rb_reset(h, p, d_mask, d_host_rand, stride, obs);
    dim3 grid((h->n + kGroupEnvs - 1) / kGroupEnvs), block(kGroupBlock);
    const bool joints = !h->cfg.is_discrete && h->cfg.action_joints, two = h->cfg.env_kind == SRLHIP_ENV_KUKA_2BUTTON;
#define SRL_TRESET(MODE)                                                                                                                       \
    if (two) hipLaunchKernelGGL((kuka_tree_reset_k<MODE, false, 2>), grid, block, 0, h->stream, p, *h->kuka, h->rng, h->stats, d_mask, d_host_rand, stride, obs); \
    else if (joints) hipLaunchKernelGGL((kuka_tree_reset_k<MODE, true, 1>), grid, block, 0, h->stream, p, *h->kuka, h->rng, h->stats, d_mask, d_host_rand, stride, obs); \
    else hipLaunchKernelGGL((kuka_tree_reset_k<MODE, false, 1>), grid, block, 0, h->stream, p, *h->kuka, h->rng, h->stats, d_mask, d_host_rand, stride, obs)
    switch (h->cfg.rng_mode) {
        case SRLHIP_RNG_HOST: SRL_TRESET(SRLHIP_RNG_HOST); break;
        case SRLHIP_RNG_PHILOX: SRL_TRESET(SRLHIP_RNG_PHILOX); break;
        default: SRL_TRESET(SRLHIP_RNG_MT19937);
    }
#undef SRL_TRESET
    SRL_HIP_CHECK(h, hipGetLastError());
    return 0;
}

int kuka_tree_settle(Handle *h, const KukaParams &p) {
    hipLaunchKernelGGL(kuka_tree_table_k, dim3(1), dim3(kGroupBlock), 0, h->stream, h->kuka->tmodel, h->kuka->ttable);
    SRL_HIP_CHECK(h, hipGetLastError());
    hipLaunchKernelGGL(kuka_tree_settle_k, dim3(1), dim3(kGroupBlock), 0, h->stream, p, *h->kuka);
    SRL_HIP_CHECK(h, hipGetLastError());
    if (h->kuka->nstarts > 0) {
        hipLaunchKernelGGL(kuka_tree_starts_k, dim3((h->kuka->nstarts + kGroupEnvs - 1) / kGroupEnvs), dim3(kGroupBlock), 0, h->stream, p, *h->kuka);
        SRL_HIP_CHECK(h, hipGetLastError());
    }
    return 0;
}

int kuka_tree_refresh(Handle *h, const KukaParams &p) {
    hipLaunchKernelGGL(kuka_tree_refresh_k, dim3((h->n + kGroupEnvs - 1) / kGroupEnvs), dim3(kGroupBlock), 0, h->stream, p, *h->kuka);
    SRL_HIP_CHECK(h, hipGetLastError());
    return 0;
}

}  // namespace srl
