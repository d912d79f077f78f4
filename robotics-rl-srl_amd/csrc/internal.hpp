// internal.hpp — handle layout shared by the C-ABI front end (api.hip) and the
// kernel translation units (mobile.hip, kuka.hip).  Not part of the ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/srlhip.h"
#include "rng.hpp"

namespace srl {

// ---- per-env random-stream state (all modes allocated lazily) --------------
struct RngState {
    Mt19937View mt;        // RNG_MT19937
    uint32_t *key;         // [2][N] Philox key (seed lo, hi) — also the action stream in every mode
    uint64_t *ctr;         // [N] Philox block counter, env stream
    uint64_t *act_ctr;     // [N] Philox block counter, synthetic-agent action stream
};

// ---- episode accounting (bench.Monitor equivalent, environments/utils.py:54)
struct EpisodeStats {
    double *ep_return;       // running
    int32_t *ep_length;      // running
    double *last_return;     // of the most recently finished episode
    int32_t *last_length;
    int32_t *n_finished;
    double *last_reward;     // reward of the last step, uncast (f64)
};

// ---- MobileRobot family: SoA state, f64 exactly as the reference's numpy ---
struct MobileState {
    double *pos_x, *pos_y;       // robot_pos[:2]            (mobile_robot_env.py:97)
    double *tgt_x, *tgt_y;       // target_pos / button_pos[0]
    double *tgt2_x, *tgt2_y;     // button_pos[1]            (2Target only)
    int32_t *counter;            // _env_step_counter
    int32_t *cur_target;         // current_target           (2Target only)
};

struct MobileParams {
    int32_t kind, is_discrete, random_target, shape_reward, auto_reset;
    int32_t n;
};

struct Handle;
struct PersistArgs;

// mobile.hip
int mobile_alloc(Handle *h);
void mobile_free(Handle *h);
int mobile_reset(Handle *h, const uint8_t *d_mask, const double *d_host_rand, float *d_obs);
int mobile_step(Handle *h, const void *d_actions, const double *d_noise, float *d_obs, float *d_rew,
                uint8_t *d_done);
int mobile_rollout(Handle *h, int T, const void *d_actions, float *d_obs, float *d_rew, uint8_t *d_done,
                   void *d_act_out);
int mobile_field(Handle *h, int field, void **dptr, size_t *elem, int *count);
int mobile_reset_rand_count(const srlhip_config &c);
int mobile_persist_blocks(Handle *h, int *capacity, uint32_t *eighths);      // persistent stepping (mobile.hip)
int mobile_persist_start(Handle *h, const void *d_actions, float *d_obs, float *d_rew, uint8_t *d_done, const struct PersistArgs &pa);

// kuka.hip
int kuka_alloc(Handle *h);
void kuka_free(Handle *h);
int kuka_reset(Handle *h, const uint8_t *d_mask, const double *d_host_rand, void *d_obs);
int kuka_step(Handle *h, const void *d_actions, const double *d_noise, void *d_obs, float *d_rew,
              uint8_t *d_done);
int kuka_rollout(Handle *h, int T, const void *d_actions, void *d_obs, float *d_rew, uint8_t *d_done,
                 void *d_act_out);
int kuka_field(Handle *h, int field, void **dptr, size_t *elem, int *count);
int kuka_reset_rand_count(const srlhip_config &c);
int kuka_refresh(Handle *h);
int kuka_uses_group_kernel(const Handle *h);
int kuka_set_model(Handle *h, const double *table138);
void kuka_default_model(double *table138);
int kuka_set_tree_model(Handle *h, const double *table510);
void kuka_default_tree_model(double *table510);
int kuka_group_probe(const double *q7_host, double *out_host, int out_doubles);
int kuka_persist_blocks(Handle *h, int *capacity);   // persistent stepping: real workgroups of the resident kernel (0: this handle has no persistent form); capacity: how many the device holds at once
int kuka_persist_start(Handle *h, const void *d_actions, float *d_obs, float *d_rew, uint8_t *d_done, const struct PersistArgs &pa);

// raster.hip
int raster_render(Handle *h, void *d_img);
// What the rasteriser sees of a Kuka handle (filled by kuka.hip::kuka_raster_view, passed BY VALUE to raster.hip's kernels — one
// definition, so that the two translation units cannot drift apart): raw plane pointers (sq / cq: [7][n] ...), gj = the five gripper
// joints of the installed full-model table (parent, frame in the parent link, axis), grip = raster_grip_k's output planes
// ([42][n]: gripper capsule end points, then the arm's joint origins); has_tm = 0 on lumped handles.
struct RasterGripJoint { double parent, xyz[3], Rj[9], axis[3]; };
struct RasterKukaView { const double *sq, *cq, *bq, *bx, *by, *bz, *b2q, *b2x, *b2y, *objs, *rb, *gsq, *gcq; RasterGripJoint gj[5]; const float *grip; int64_t n; int32_t two, rand_objects, has_tm; };

struct KukaState;   // defined in kuka.hip

// Persistent stepping (srlhip_set_persistent): the per-step API without a launch per step.  ONE launch of the rollout kernel stays
// resident — every wavefront keeps its envs' state in registers — and takes its steps from the host through mapped memory: the host
// writes the actions, then a new sequence number; workgroup 0 polls that word over PCIe and relays it through a device-memory word
// (one poller on the bus, not 1024); every wavefront steps, writes its outputs to the mapped planes and its own `done` word.  The kernel
// PARKS (writes the state back and exits) when told to (any other API call on the handle) or when no step arrived for park_us.
struct PersistArgs {
    const uint32_t *seq, *stop;     // host-written (mapped, coherent): sequence number of the newest step; 1 = park now
    uint32_t *parked;               // device-written: workgroup 0 decided to park (the host must synchronise and relaunch)
    uint32_t *done;                 // device-written [8]: sequence number of the last step that eighth of the workgroups finished
    uint32_t *relay;                // device memory [8 x stride]: workgroup 0's token for the others (a sequence number, or kPersistPark)
    uint32_t *count;                // device memory [8 x stride]: arrivals per eighth of the workgroups, never reset while resident
    uint32_t *ctrl;                 // device memory: [0] workgroups registered | XCD mismatches << 16, [stride] the start barrier's verdict
    const uint32_t *stage;          // device memory: the staging copy of the step's output planes (same layout as the host's), dword view
    uint32_t *host_out;             // the host's mapped output planes, dword view; reward / done planes start rew_dw / done_dw dwords in
    uint32_t rew_dw, done_dw;
    uint32_t start_seq, spin_limit;
    uint32_t force_staged;          // SRLHIP_PERSIST_STAGED=1: the staging copy + copier even where the direct form is valid (tests run both)
};
constexpr int kPersistWordStride = 64;         // (uint32 words: 256 bytes between two relay / counter words)
constexpr uint32_t kPersistPark = 0xffffffffu;

struct Handle {
    srlhip_config cfg;
    int n;
    hipStream_t stream;
    hipEvent_t ev_begin, ev_end;
    std::string err;
    RngState rng;
    EpisodeStats stats;
    MobileState mobile;
    // read-only snapshot of everything mobile_rollout_ep_k's segment lanes read from the live state (taken by a copy kernel
    // right before the launch): a lane may start after the lane that writes its env's final state has retired
    MobileState mobile_snap = {};
    uint64_t *snap_ctr = nullptr;
    double *snap_ep_return = nullptr;
    int32_t *snap_ep_length = nullptr;
    // ... and its twin (round 5): the lane that leaves an env's final state also writes it into the OTHER snapshot set, which the next
    // rollout launch reads — back-to-back rollouts need no mobile_snapshot_k launch (4.8 us of a 52 us rollout).  snap_cur = the set the
    // next launch reads; snap_valid = that set equals the live state (cleared by everything else that touches the state)
    MobileState mobile_snap2 = {};
    uint64_t *snap2_ctr = nullptr, *snap2_actr = nullptr;
    double *snap2_ep_return = nullptr;
    int32_t *snap2_ep_length = nullptr;
    int snap_cur = 0;
    bool snap_valid = false;
    // synthetic-agent rollouts of the MobileRobot family: the [T][N] action plane of the NEXT rollout is drawn by spare workgroups
    // of the current rollout's launch (mobile_rollout_ep_k), from the action-stream counters in the snapshot; two planes in turn
    uint64_t *snap_actr = nullptr;
    void *act_plane[2] = {nullptr, nullptr};
    size_t act_plane_sz[2] = {0, 0};
    bool prefetch_valid = false;
    int prefetch_T = 0, prefetch_buf = 0;
    KukaState *kuka;
    double kuka_tmodel_host[510] = {0};   // host copy of the installed full-model table (srlhip_kuka_tree_model): the rasteriser's gripper joint frames
    std::vector<void *> allocs;      // everything hipMalloc'ed for this handle
    // staging buffers for io_device == 0
    void *st_actions, *st_noise, *st_obs, *st_rew, *st_done, *st_mask, *st_rand;
    size_t st_actions_sz, st_noise_sz, st_obs_sz, st_rew_sz, st_done_sz, st_mask_sz, st_rand_sz;
    // rasteriser: per fixed camera, the unit ray of every pixel + depth / colour of the static scenery behind it
    // (built lazily on the first render; lives in `allocs`)
    float4 *raster_rays[2];
    uint32_t *raster_bg[2];
    float *raster_grip = nullptr;         // [42][n] full Kuka model: the gripper capsules' end points, then the arm's joint origins (raster_grip_k, refreshed by every render)
    void *pin_in, *pin_out;          // pinned host bounce buffers of srlhip_step (host-pointer mode)
    // host-pointer handles (io_device = 0): stats.last_return / last_length live in ONE mapped pinned host block ([n] f64, then [n] i32)
    // that the kernels address directly, so the per-step API reads an ended episode's (r, l) without a device copy (srlhip_episode_records)
    void *ep_host = nullptr;
    size_t pin_in_sz, pin_out_sz;
    bool step_pending = false;       // srlhip_step_async enqueued a step that srlhip_step_wait has not collected yet
    // persistent stepping (PersistArgs): the mapped control block [seq, stop, parked, pad..., done[blocks]], the device relay word
    bool persist_on = false, persist_running = false, persist_step = false;
    void *persist_host = nullptr;
    uint32_t *persist_relay = nullptr;
    void *persist_stage = nullptr;
    uint32_t persist_seq = 0, persist_blocks = 0, persist_park_us = 2000;
    // early completion signal of single-step launches (host_step_begin / host_step_finish; kuka_tree_launch arms it where the kernel supports it)
    const PersistArgs *step_signal = nullptr;
    bool step_signal_armed = false, signal_wait = false;
    uint32_t signal_seq = 0, signal_steps = 0, signal_fallbacks = 0;
    uint32_t signal_eighths = 0;     // which of the 8 `done` words the armed launch will write (set by the launcher)
    uint32_t persist_eighths = 0;    // which of the 8 `done` words the resident kernel writes
    int persist_reserved = 0;        // workgroups this handle holds in the per-device residency tally (api.hip)

    int fail(int code, const std::string &msg) { err = msg; return code; }
    template <class T>
    int dalloc(T **p, size_t count) {
        void *q = nullptr;
        hipError_t e = hipMalloc(&q, count * sizeof(T) > 0 ? count * sizeof(T) : 16);
        if (e != hipSuccess) return fail(SRLHIP_ENOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
        e = hipMemsetAsync(q, 0, count * sizeof(T), stream);
        if (e != hipSuccess) return fail(SRLHIP_EHIP, std::string("hipMemset: ") + hipGetErrorString(e));
        allocs.push_back(q);
        *p = static_cast<T *>(q);
        return 0;
    }
};

#define SRL_HIP_CHECK(h, expr)                                                                     \
    do {                                                                                           \
        hipError_t e__ = (expr);                                                                   \
        if (e__ != hipSuccess)                                                                     \
            return (h)->fail(SRLHIP_EHIP, std::string(#expr ": ") + hipGetErrorString(e__));       \
    } while (0)

inline int obs_dim_of(const srlhip_config &c) {
    switch (c.env_kind) {
        case SRLHIP_ENV_MOBILE_1D: return 1;
        case SRLHIP_ENV_MOBILE: case SRLHIP_ENV_MOBILE_2TARGET: case SRLHIP_ENV_MOBILE_LINE: return 2;
        case SRLHIP_ENV_KUKA_BUTTON: case SRLHIP_ENV_KUKA_MOVING: case SRLHIP_ENV_KUKA_2BUTTON: case SRLHIP_ENV_KUKA_RAND:
            return c.obs_mode == SRLHIP_OBS_JOINTS ? 14 : c.obs_mode == SRLHIP_OBS_JOINTS_POSITION ? 17 : 3;
    }
    return 0;
}
inline int num_actions_of(const srlhip_config &c) {
    if (!c.is_discrete) return 0;
    switch (c.env_kind) {
        case SRLHIP_ENV_MOBILE_1D: return 2;
        case SRLHIP_ENV_KUKA_BUTTON: case SRLHIP_ENV_KUKA_MOVING: case SRLHIP_ENV_KUKA_2BUTTON: case SRLHIP_ENV_KUKA_RAND: return 6;
        default: return 4;
    }
}
inline int action_dim_of(const srlhip_config &c) {
    if (c.is_discrete) return 1;
    if (c.env_kind >= SRLHIP_ENV_KUKA_BUTTON) return c.action_joints ? 7 : 3;
    return 2;
}

}  // namespace srl
