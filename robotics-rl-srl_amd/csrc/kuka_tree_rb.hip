// kuka_tree_rb.hip — KukaRandButtonGymEnv on the full-model lane-group stepper: the RB = 1 instantiations of the rollout / reset kernels
// (kuka_tree_kernels.hpp), i.e. the arm + gripper tree with the env's eleven free bodies (ten distractors + the kicked ball,
// kuka_rand_button_gym_env.py:59-71,111-125) on lanes 0..10 — kuka_tree.hpp, free-body section.  A translation unit of its own so that it
// compiles beside kuka_tree.hip.
#include "kuka_tree_kernels.hpp"

namespace srl {
using namespace kuka;

#define SRL_TREE_RB_GO(MODE, J, G) hipLaunchKernelGGL((kuka_tree_rollout_k<MODE, J, G, 1, 1>), grid, block, 0, h->stream, p, *h->kuka, h->rng, h->stats, T, d_actions, d_noise, obs, d_rew, d_done, d_act_out, G ? sig : PersistArgs{})
#define SRL_TREE_RB_MODE(MODE)                                      \
    if (joints && d_actions) SRL_TREE_RB_GO(MODE, true, true);      \
    else if (joints) SRL_TREE_RB_GO(MODE, true, false);             \
    else if (d_actions) SRL_TREE_RB_GO(MODE, false, true);          \
    else SRL_TREE_RB_GO(MODE, false, false);
int kuka_tree_rb_launch(Handle *h, const KukaParams &p, int T, const void *d_actions, const double *d_noise, float *obs, float *d_rew,
                        uint8_t *d_done, void *d_act_out) {
    dim3 grid(((h->n + kGroupEnvs - 1) / kGroupEnvs + 7) / 8 * 8), block(kGroupBlock);      // a multiple of 8: the rollout kernel maps blocks to envs XCD by XCD
    const bool joints = !h->cfg.is_discrete && h->cfg.action_joints;
    PersistArgs sig{};                          // the early completion signal of a single-step launch (kuka_tree.hip, api.hip)
    if (h->step_signal && T == 1 && d_actions) {
        sig = *h->step_signal; h->step_signal_armed = true;
        const uint32_t blocks = (uint32_t)(h->n + kGroupEnvs - 1) / kGroupEnvs, per = (blocks + 7) / 8;
        h->signal_eighths = 0;
        for (uint32_t g = 0; g < 8 && g * per < blocks; g++) h->signal_eighths |= 1u << g;
    }
    switch (h->cfg.rng_mode) {
        case SRLHIP_RNG_PHILOX: SRL_TREE_RB_MODE(SRLHIP_RNG_PHILOX) break;
        case SRLHIP_RNG_MT19937: SRL_TREE_RB_MODE(SRLHIP_RNG_MT19937) break;
        default: SRL_TREE_RB_MODE(SRLHIP_RNG_HOST)
    }
    SRL_HIP_CHECK(h, hipGetLastError());
    return 0;
}

int kuka_tree_rb_reset(Handle *h, const KukaParams &p, const uint8_t *d_mask, const double *d_host_rand, int stride, float *obs) {
    dim3 grid((h->n + kGroupEnvs - 1) / kGroupEnvs), block(kGroupBlock);
    const bool joints = !h->cfg.is_discrete && h->cfg.action_joints;
#define SRL_TRESET_RB(MODE)                                                                                                                    \
    if (joints) hipLaunchKernelGGL((kuka_tree_reset_k<MODE, true, 1, 1>), grid, block, 0, h->stream, p, *h->kuka, h->rng, h->stats, d_mask, d_host_rand, stride, obs); \
    else hipLaunchKernelGGL((kuka_tree_reset_k<MODE, false, 1, 1>), grid, block, 0, h->stream, p, *h->kuka, h->rng, h->stats, d_mask, d_host_rand, stride, obs)
    switch (h->cfg.rng_mode) {
        case SRLHIP_RNG_HOST: SRL_TRESET_RB(SRLHIP_RNG_HOST); break;
        case SRLHIP_RNG_PHILOX: SRL_TRESET_RB(SRLHIP_RNG_PHILOX); break;
        default: SRL_TRESET_RB(SRLHIP_RNG_MT19937);
    }
#undef SRL_TRESET_RB
    SRL_HIP_CHECK(h, hipGetLastError());
    return 0;
}

}  // namespace srl
