// kuka_device.hpp — what the Kuka translation units share: state planes in HBM, kernel parameters, the device-side random
// stream adaptors.  kuka.hip holds the lane-per-env kernels and the host plumbing, kuka_group.hip / kuka_group_cm.hip the
// lane-group kernels (baked model / runtime model table): three translation units so that they compile in parallel.
#pragma once
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "internal.hpp"
#include "kuka_env.hpp"
#include "kuka_tree_model.hpp"

namespace srl {


using namespace kuka;

constexpr int kWave = 64;
constexpr int NDBL = 67, NINT = 10;
constexpr int kGroupKernelMaxEnvs = 12288;     // batches up to this size are stepped by the lane-group kernel (measured crossover, profiles/r02_nsweep_kuka.jsonl)

// SoA planes (doubles): q7 qd7 sq7 cq7 ee3 bq bqd bx by bpos3 grip3
enum { D_Q = 0, D_QD = 7, D_SQ = 14, D_CQ = 21, D_EE = 28, D_BQ = 31, D_BQD = 32, D_BX = 33, D_BY = 34, D_BPOS = 35, D_GRIP = 38, D_BZ = 41, D_BSPEED = 42,
       D_B2Q = 43, D_B2QD = 44, D_B2X = 45, D_B2Y = 46,        // second button (Kuka2ButtonGymEnv)
       D_GQ = 47, D_GQD = 52, D_GSQ = 57, D_GCQ = 62 };        // full model: gripper DoFs 7..11 (q, qd, sin, cos)
// plane of joint lane l of the full model: arm joints in the q7 planes, gripper joints behind them
__host__ __device__ inline int tree_plane(int base_arm, int base_gripper, int l) { return l < ND ? base_arm + l : base_gripper + (l - ND); }
// SoA planes (int32): motor_on contact_button contact_table counter n_contacts n_outside terminated
enum { I_MOTOR = 0, I_CB = 1, I_CT = 2, I_COUNTER = 3, I_NCONTACT = 4, I_NOUT = 5, I_TERM = 6, I_GOAL = 7, I_NCONTACT2 = 8,
       I_IKX = 9 };          // full model: IK conditioning flag (bit 0, sticky per episode) + flagged env-steps << 1 (Env::ikx)

struct KukaState {
    double *d;          // [NDBL][n]
    int32_t *i;         // [NINT][n]
    double *rows;       // [SC_ROWS_TOTAL][n]  generic constraint rows (global scratch, L2-resident)
    double *objs;       // [30][n]  KukaRandButton distractor objects (x, y, present) x 10
    double *settled;    // [kStartDoubles]
    double *starts;     // [nstarts][kStartDoubles]
    int32_t nstarts;
    Model *model;       // runtime model table (device copy); the baked one unless srlhip_set_kuka_model() installed another
    int32_t custom_model;
    // full model (cfg.kuka_model = SRLHIP_KUKA_MODEL_FULL, kuka_tree.hpp): its own table and start states
    int32_t full;
    TreeModel *tmodel;
    double *ttable;     // [tree::kLaneTableDoubles] per-lane constants derived from tmodel (kuka_tree_table_k)
    double *tsettled;   // [kTreeStartDoubles]
    double *tstarts;    // [nstarts][kTreeStartDoubles]
    double *rb;         // [66][n]  KukaRandButton free bodies, full model: plane 6 k + c = (x y z vx vy vz)[c] of body k (0..9 distractors, 10 the ball)
};


struct KukaParams { kuka::Cfg cfg; int32_t n; };

struct DevMt { Mt19937 m;
    __device__ double double01() { return m.double01(); } __device__ double uniform(double a, double b) { return m.uniform(a, b); }
    __device__ double normal(double a, double b) { return m.normal(a, b); } __device__ uint32_t bounded(uint32_t r) { return m.bounded(r); } };
struct DevPhilox { Philox p;
    __device__ double double01() { return p.double01(); } __device__ double uniform(double a, double b) { return p.uniform(a, b); }
    __device__ double normal(double a, double b) { return p.normal(a, b); } __device__ uint32_t bounded(uint32_t r) { return p.bounded(r); } };
template <int MODE> struct KRng;
template <> struct KRng<SRLHIP_RNG_HOST> { using type = HostDraws; };
template <> struct KRng<SRLHIP_RNG_PHILOX> { using type = DevPhilox; };
template <> struct KRng<SRLHIP_RNG_MT19937> { using type = DevMt; };

template <int MODE>
__device__ __forceinline__ void krng_load(typename KRng<MODE>::type &r, const RngState &rs, int e, int n, const double *draws) {
    if constexpr (MODE == SRLHIP_RNG_HOST) { r.v = draws; r.i = 0; }
    else if constexpr (MODE == SRLHIP_RNG_PHILOX) { r.p.k0 = rs.key[e]; r.p.k1 = rs.key[n + e]; r.p.ctr = rs.ctr[e]; r.p.stream = 0; }
    else r.m.load(rs.mt, e);
}
template <int MODE>
__device__ __forceinline__ void krng_store(const typename KRng<MODE>::type &r, const RngState &rs, int e) {
    if constexpr (MODE == SRLHIP_RNG_PHILOX) rs.ctr[e] = r.p.ctr;
    else if constexpr (MODE == SRLHIP_RNG_MT19937) r.m.store(rs.mt, e);
}


inline KukaParams params_of(const Handle *h) {

    KukaParams p;
    const srlhip_config &c = h->cfg;
    p.cfg.random_target = c.random_target; p.cfg.force_down = c.force_down; p.cfg.shape_reward = c.shape_reward;
    p.cfg.action_repeat = c.action_repeat; p.cfg.is_discrete = c.is_discrete; p.cfg.action_joints = c.action_joints;
    p.cfg.obs_mode = c.obs_mode; p.cfg.auto_reset = c.auto_reset; p.cfg.max_distance = c.max_distance;
    p.cfg.moving = c.env_kind == SRLHIP_ENV_KUKA_MOVING ? 1 : 0;
    p.cfg.two = c.env_kind == SRLHIP_ENV_KUKA_2BUTTON ? 1 : 0;
    p.cfg.rand_objects = c.env_kind == SRLHIP_ENV_KUKA_RAND ? 1 : 0;
    p.cfg.info_bits = c.info_bits;
    p.cfg.max_steps = p.cfg.moving ? 1500 : p.cfg.two ? kMaxSteps2Button : kMaxSteps;
    p.n = h->n;
    return p;
}


// lane-group kernels: launchers (kuka_group.hip: baked model; kuka_group_cm.hip: runtime model table)
constexpr int kGroupBlock = 64;
constexpr int kGroupEnvs = kGroupBlock / 16;
int kuka_group_launch_baked(Handle *h, const KukaParams &p, int T, const void *d_actions, const double *d_noise, float *obs, float *d_rew,
                            uint8_t *d_done, void *d_act_out);
int kuka_group_launch_table(Handle *h, const KukaParams &p, int T, const void *d_actions, const double *d_noise, float *obs, float *d_rew,
                            uint8_t *d_done, void *d_act_out);
int kuka_group_reset_table(Handle *h, const KukaParams &p, const uint8_t *d_mask, const double *d_host_rand, int stride, float *obs);
int kuka_group_settle_table(Handle *h, const KukaParams &p);
// full-model lane-group kernels (kuka_tree.hip)
// persistent stepping: 1 when this handle's configuration has a persistent instantiation and its grid is co-resident on the device
int kuka_tree_persist_blocks(Handle *h, int *capacity);       // > 0: the number of real workgroups; 0: not supported.  capacity: workgroups of that kernel the device holds at once
int kuka_tree_persist_launch(Handle *h, const KukaParams &p, const void *d_actions, float *obs, float *d_rew, uint8_t *d_done, const PersistArgs &pa);
int kuka_tree_launch(Handle *h, const KukaParams &p, int T, const void *d_actions, const double *d_noise, float *obs, float *d_rew,
                     uint8_t *d_done, void *d_act_out);
int kuka_tree_reset(Handle *h, const KukaParams &p, const uint8_t *d_mask, const double *d_host_rand, int stride, float *obs);
// two wavefronts per SIMD (default from 65536 envs, SRLHIP_KUKA_OCC forces either): kuka_tree_occ.hip
int kuka_tree_occ_launch(Handle *h, const KukaParams &p, int T, const void *d_actions, const double *d_noise, float *obs, float *d_rew,
                         uint8_t *d_done, void *d_act_out);
// KukaRandButtonGymEnv (free bodies): kuka_tree_rb.hip
int kuka_tree_rb_launch(Handle *h, const KukaParams &p, int T, const void *d_actions, const double *d_noise, float *obs, float *d_rew,
                        uint8_t *d_done, void *d_act_out);
int kuka_tree_rb_reset(Handle *h, const KukaParams &p, const uint8_t *d_mask, const double *d_host_rand, int stride, float *obs);
int kuka_tree_settle(Handle *h, const KukaParams &p);     // settled state + start-state table of the installed tree model
int kuka_tree_refresh(Handle *h, const KukaParams &p);

}  // namespace srl
