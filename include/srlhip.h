/*
 * srlhip.h — C-ABI of libsrlhip, the MI355X (gfx950) vectorised env stepper.
 *
 * Drop-in boundary for the env-step hot path of araffin/robotics-rl-srl.
 * The reference has no native ABI of its own (pure Python over the
 * third-party pybullet wheel); the seam it exposes is Python-level:
 *   - per-env gym surface      environments/srl_env.py:5-102
 *   - vec-env assembly         rl_baselines/utils.py:194-229 (createEnvs)
 *   - dataset generator loop   environments/dataset_generator.py:38-117
 * Each entry point below names the reference interface it replaces.
 * The Python host side (robotics-rl-srl_amd/srlhip) binds these with ctypes;
 * INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *   - every function returns 0 on success, a negative SRLHIP_E* code otherwise;
 *     srlhip_last_error() gives the message (per handle; NULL handle ->
 *     thread-local message of the last failed srlhip_create).
 *   - the library owns all device memory (structure-of-arrays, one array per
 *     field, env index fastest).  Pointer arguments are borrowed for the call.
 *     cfg.io_device = 0: they are HOST pointers (the call copies and
 *     synchronises);  cfg.io_device = 1: they are DEVICE pointers on
 *     cfg.device_id and the call only enqueues work on the handle's stream
 *     (srlhip_sync() or the returned stream to order against it).
 *   - one HIP stream per handle; a handle is not thread-safe; distinct handles
 *     are independent (no global mutable state) -> one handle per GPU.
 *   - "env id" i of a handle is GLOBAL env cfg.first_env_id + i; every random
 *     stream is derived from the global id so results are invariant to how
 *     envs are sharded across GPUs (environments/utils.py:52 seeds seed+rank).
 */
#ifndef SRLHIP_H
#define SRLHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SRLHIP_ABI_VERSION 5   /* 2 (round 4): srlhip_kuka_tree_model grew its solver section (506 -> 510 doubles), SRLHIP_F_KUKA_BODIES;
                                 * 3 (round 5): SRLHIP_F_KUKA_IK_CROSSED;
                                 * 4 (round 6): srlhip_step_async / srlhip_step_wait / srlhip_step_pending, srlhip_config.info_bits
                                 *              (was reserved0);
                                 * 5 (round 6): srlhip_set_persistent */

/* ---- error codes ------------------------------------------------------- */
#define SRLHIP_OK            0
#define SRLHIP_EINVAL      (-22)  /* bad argument / unsupported combination     */
#define SRLHIP_ENOMEM      (-12)
#define SRLHIP_EHIP        (-5)   /* HIP runtime error (no device, launch fail) */
#define SRLHIP_ENOTSUP     (-95)  /* e.g. continuous actions on Mobile2Target   */

/* ---- env kinds: environments/registry.py:41-53 -------------------------- */
#define SRLHIP_ENV_MOBILE          0  /* MobileRobotGymEnv-v0            */
#define SRLHIP_ENV_MOBILE_1D       1  /* MobileRobot1DGymEnv-v0          */
#define SRLHIP_ENV_MOBILE_2TARGET  2  /* MobileRobot2TargetGymEnv-v0     */
#define SRLHIP_ENV_MOBILE_LINE     3  /* MobileRobotLineTargetGymEnv-v0  */
#define SRLHIP_ENV_KUKA_BUTTON     4  /* KukaButtonGymEnv-v0             */
#define SRLHIP_ENV_KUKA_MOVING     5  /* KukaMovingButtonGymEnv-v0 (kuka_moving_button_gym_env.py) */
#define SRLHIP_ENV_KUKA_2BUTTON    6  /* Kuka2ButtonGymEnv-v0 (kuka_2button_gym_env.py): two buttons pressed in order */
#define SRLHIP_ENV_KUKA_RAND       7  /* KukaRandButtonGymEnv-v0 (kuka_rand_button_gym_env.py:59-71,111-125): the ten dropped objects and the kicked ball are
                                        * free bodies the arm can push (full model: proxy shapes, table contact + friction, arm-sphere contacts —
                                        * srlhip_get_state KUKA_BODIES); scenery at rest with the lumped model */
#define SRLHIP_ENV_LAST            SRLHIP_ENV_KUKA_RAND

/* ---- observation modes: kuka_button_gym_env.py:162-173 ------------------ */
#define SRLHIP_OBS_GROUND_TRUTH     0  /* f32[obs_dim] relative position        */
#define SRLHIP_OBS_JOINTS           1  /* Kuka only: 14 constant joint values   */
#define SRLHIP_OBS_JOINTS_POSITION  2  /* Kuka only: 3 + 14                      */
#define SRLHIP_OBS_RAW_PIXELS       3  /* u8[H,W,3(6)] from the tile rasteriser  */

/* ---- Kuka body model: what loadSDF("kuka_iiwa/kuka_with_gripper2.sdf") (kuka.py:60) becomes ------------------
 * FULL   : the 12-DoF tree — arm J0..J6, gripper_to_arm, two fingers, two finger tips — every joint driven by the
 *          POSITION_CONTROL motor the reference commands each step (kuka.py:167-187: arm 200 N·m / 0.35 rad/s, joint 7
 *          200, fingers 2 / 2.5, tips 2), contact spheres on links 5..11, one friction row per contact.  Stepped by the
 *          tree lane-group kernel (csrc/kuka_tree.hpp) at every batch size.  The default of every Kuka env
 *          (Kuka2ButtonGymEnv: the kernel's two-button form — second glider, its motor / stop rows, its cap and base).
 * LUMPED : rounds 1-2 — the gripper welded to link_7 (7 DoF), six gripper spheres, frictionless contacts.  Kept as the
 *          cheaper approximation (profiles/r03_kuka_model_gap.json measures what it costs: 0.02 rad on the arm joints,
 *          different reward / done planes). */
#define SRLHIP_KUKA_MODEL_LUMPED 0
#define SRLHIP_KUKA_MODEL_FULL   1

/* ---- random-number modes ------------------------------------------------- */
#define SRLHIP_RNG_HOST     0  /* caller supplies every random draw (parity harness)        */
#define SRLHIP_RNG_PHILOX   1  /* device Philox4x32-10 keyed (seed, global env id): throughput */
#define SRLHIP_RNG_MT19937  2  /* device-resident numpy RandomState per env: reproduces the
                                  reference's env.np_random stream (srl_env.py:71-78)       */

typedef struct srlhip_env *srlhip_handle;

typedef struct srlhip_config {
    int32_t struct_size;      /* = sizeof(srlhip_config), ABI check                        */
    int32_t env_kind;         /* SRLHIP_ENV_*                                              */
    int32_t num_envs;         /* envs owned by this handle (this GPU's shard)              */
    int32_t device_id;        /* HIP device ordinal                                        */
    int32_t first_env_id;     /* global id of env 0 (sharding offset)                      */
    int32_t is_discrete;      /* ctor kwarg is_discrete   (mobile_robot_env.py:61)         */
    int32_t random_target;    /* ctor kwarg random_target                                  */
    int32_t force_down;       /* ctor kwarg force_down    (Kuka)                           */
    int32_t shape_reward;     /* ctor kwarg shape_reward                                   */
    int32_t action_repeat;    /* ctor kwarg action_repeat (Kuka, >=1)                      */
    int32_t action_joints;    /* ctor kwarg action_joints (Kuka)                           */
    int32_t obs_mode;         /* SRLHIP_OBS_*  (ctor kwarg srl_model)                      */
    int32_t img_h, img_w;     /* raw_pixels size (reference: 224x224)                      */
    int32_t multi_view;       /* second camera stacked behind the first (6-channel image): ctor kwarg
                                 multi_view (Kuka, kuka_button_gym_env.py:401-417) / fpv (MobileRobot,
                                 mobile_robot_env.py:313-332: the camera riding on the robot)       */
    int32_t rng_mode;         /* SRLHIP_RNG_*                                              */
    int32_t auto_reset;       /* 1: step() resets finished envs itself and returns the first
                                 observation of the next episode (SB VecEnv semantics,
                                 rl_baselines/utils.py:216-220); needs rng_mode != HOST     */
    int32_t io_device;        /* 0 host pointers, 1 device pointers (see Conventions)      */
    int32_t kuka_model;       /* SRLHIP_KUKA_MODEL_* (Kuka envs; srlhip_default_config picks FULL where it exists) */
    int32_t info_bits;        /* 0: done_out bytes are 0 / 1.  1 (full-model Kuka handles; ignored elsewhere): bit 1 of every
                                 done_out byte = the env-step ran under the IK conditioning flag (SRLHIP_F_KUKA_IK_CROSSED bit 0,
                                 sampled BEFORE an auto-reset clears it) — the per-step `infos` of the VecEnv surface    */
    int64_t seed0;            /* base seed: env i is seeded seed0 + first_env_id + i       */
    double  max_distance;     /* ctor kwarg max_distance                                   */
} srlhip_config;

/* Fill *cfg with the reference's ctor defaults for env_kind
 * (kuka_button_gym_env.py:78-81, mobile_robot_env.py:61-64). */
int srlhip_default_config(int32_t env_kind, srlhip_config *cfg);

/* Replaces: env construction in makeEnv/_make (environments/utils.py:36-95),
 * one call for the whole batch instead of one process per env. Seeds every env
 * with seed0 + global id, but does not reset. */
int srlhip_create(const srlhip_config *cfg, srlhip_handle *out);
int srlhip_destroy(srlhip_handle h);

/* Shapes implied by the config (observation_space / action_space of the env). */
int srlhip_obs_dim(srlhip_handle h);          /* floats per env (ground truth modes)   */
int srlhip_obs_bytes(srlhip_handle h);        /* bytes per env of one observation      */
int srlhip_action_dim(srlhip_handle h);       /* 1 (discrete) | 2 | 3 | 7              */
int srlhip_num_actions(srlhip_handle h);      /* Discrete(n): n ; 0 if continuous      */

/* Replaces SRLGymEnv.seed (srl_env.py:71-78) for the envs selected by mask
 * (NULL = all).  seeds[i] is the integer the reference would pass to
 * env.seed(); MT19937 mode hashes it exactly like gym (sha512 -> init_by_array).
 * mask/seeds are always HOST pointers. */
int srlhip_seed(srlhip_handle h, const uint8_t *mask, const int64_t *seeds);

/* Replaces <Env>.reset() (mobile_robot_env.py:159-222, kuka_button_gym_env.py:214-281)
 * for the envs selected by mask (NULL = all; HOST pointer).
 * host_rand: RNG_HOST only — [num_envs][srlhip_reset_rand_count()] doubles, the
 *            values the reference's np_random calls would return, in draw order.
 * obs_out:   [num_envs][obs] — rows of unselected envs are left untouched. */
int srlhip_reset(srlhip_handle h, const uint8_t *mask, const double *host_rand, void *obs_out);
int srlhip_reset_rand_count(srlhip_handle h);

/* Replaces VecEnv.step / <Env>.step (mobile_robot_env.py:235-280,
 * kuka_button_gym_env.py:293-368) for the whole batch.
 * actions: discrete -> int32[num_envs], value -1 == the reference's `None`
 *          action (kuka_button_gym_env.py:295-299);  continuous -> float[num_envs][adim].
 * host_noise: RNG_HOST only — double[num_envs], value of np_random.normal(...) per env.
 * obs_out [num_envs][obs], reward_out float[num_envs], done_out uint8[num_envs].
 * With cfg.auto_reset the observation of a finished env is the first one of its
 * next episode. */
int srlhip_step(srlhip_handle h, const void *actions, const double *host_noise,
                void *obs_out, float *reward_out, uint8_t *done_out);

/* srlhip_step in two halves, for HOST-pointer handles (cfg.io_device = 0): what SubprocVecEnv.step_async / step_wait are to the
 * reference (rl_baselines/utils.py:213-220 builds the SubprocVecEnv whose step_async only SENDS the actions to the workers and whose
 * step_wait collects them).  srlhip_step_async validates the inputs, copies them into the handle's pinned block and ENQUEUES the
 * step on the handle's stream: it returns before the GPU has run it, so a caller holding G handles (one per GPU — the sharded
 * HipVecEnv) has all G devices stepping at once, and a single-handle caller overlaps its own work with the step.
 * srlhip_step_wait waits for that step and copies its planes out (any pointer may be NULL); with cfg.auto_reset the observation of
 * a finished env is the first one of its next episode.  One step may be pending per handle: a second srlhip_step_async, or a
 * srlhip_step, before the wait returns SRLHIP_EINVAL; srlhip_step_wait without a pending step too.  srlhip_step ==
 * srlhip_step_async + srlhip_step_wait.  srlhip_step_pending: 1 between the two, else 0. */
int srlhip_step_async(srlhip_handle h, const void *actions, const double *host_noise);
int srlhip_step_wait(srlhip_handle h, void *obs_out, float *reward_out, uint8_t *done_out);
int srlhip_step_pending(srlhip_handle h);
/* (Zero-copy steps of the full-model Kuka kernels and of the MobileRobot family do not wait for the kernel's END: the kernel reports the step's outputs per
 * eighth of its grid — one XCD each, checked per launch — after one L2 write-back, and srlhip_step / srlhip_step_wait poll
 * those words; the stream synchronisation remains the fallback.  SRLHIP_STEP_SIGNAL=0 switches the signal off.  What the call
 * returns — observations, rewards, dones, Monitor's records — is complete; the kernel's exit stores of the STATE planes may still be
 * in flight on the handle's stream: every entry point of this library is ordered behind them, a caller that reads
 * srlhip_device_ptr() planes of a host-pointer handle from a stream of its own calls srlhip_sync() first.) */

/* Persistent stepping (opt-in; host-pointer handles of the MobileRobot family, KukaButtonGymEnv, KukaMovingButtonGymEnv and
 * Kuka2ButtonGymEnv — any action mode and any observation mode but raw pixels — on a device RNG mode, whose wavefronts are all resident at once: up to
 * 4096 envs on an MI355X; SRLHIP_ENOTSUP otherwise): the step pair WITHOUT a kernel launch per step.  One launch of the rollout kernel stays on
 * the device with every env's state in registers.  srlhip_step_async writes the actions and a sequence number into mapped
 * memory; workgroup 0 polls that word over PCIe and relays it through device memory; every wavefront has already run everything
 * in front of the action (dynamics, collision detection, on a contact step the solver's setup) while it waited, finishes the step and writes its
 * outputs to the host's mapped planes with plain stores — they stay in its XCD's L2; the last wavefront of each eighth of the
 * grid (= one XCD, verified at launch through HW_REG_XCC_ID) to finish writes that L2 back with ONE system-scope release and
 * writes the eighth's `done` word; srlhip_step_wait polls those 8 words.  (Where an eighth does not sit on one XCD, or with
 * SRLHIP_PERSIST_STAGED=1, the outputs go through a staging copy in device memory that the eighth's last wavefront copies out.)  What a per-step launch pays every time — the launch itself, the
 * 5.6 KB model table, ~40 state planes, the generators, forward kinematics, the stream synchronisation — is paid once: measured
 * HipVecEnv.step 82 -> 62 us at 4096 envs, 61 -> 44 at 256, 39 -> 28 at 16 (launching path -> persistent; MobileRobot: 19 -> 13 us
 * and 13 -> 6 us at 16 envs).  Same kernel
 * code, same arithmetic: results are those of the launching path bit for bit (tests/test_gpu_persistent_step.py).
 * The kernel PARKS (writes the state back and exits) when any other entry point touches the handle, and by itself when no step
 * arrived for park_us microseconds (<= 0: 2000) — the next step restarts it, at the cost of a launch.  While it is resident it
 * holds one wavefront slot and 40 KB of LDS on every SIMD it uses (all of them at 4096 envs): use it when the policy runs on
 * the host or on another device (a random agent, ARS / CMA-ES with a numpy policy, a policy server) and answers within park_us.
 * A workgroup that could not become resident shows up as a step that never completes: srlhip_step_wait then parks the kernel
 * and fails with SRLHIP_EHIP after ~20 s instead of hanging.  on = 0 switches back to launches. */
int srlhip_set_persistent(srlhip_handle h, int32_t on, int32_t park_us);

/* Fused rollout: T consecutive steps with auto-reset, outputs streamed as
 * [T][num_envs] planes.  Replaces the random-agent hot loop
 * (rl_baselines/random_agent.py:35-42, dataset_generator.py:91-105).
 * actions_TN: int32/float [T][num_envs]([adim]) or NULL -> uniform random
 * actions drawn on the device (returned in act_out_TN if not NULL).
 * Requires cfg.auto_reset and rng_mode != HOST.  Any output may be NULL. */
int srlhip_rollout(srlhip_handle h, int32_t T, const void *actions_TN,
                   void *obs_TN, float *reward_TN, uint8_t *done_TN, void *act_out_TN);

/* Raw state access (checkpoint / parity): field ids below; arrays are
 * [num_envs] (or [k][num_envs] for vector fields) of the field's own type.
 * Always HOST pointers. */
#define SRLHIP_F_POS_X          0   /* f64 */
#define SRLHIP_F_POS_Y          1   /* f64 */
#define SRLHIP_F_TARGET_X       2   /* f64 (current target for 2Target)      */
#define SRLHIP_F_TARGET_Y       3   /* f64 */
#define SRLHIP_F_STEP_COUNT     4   /* i32 _env_step_counter                 */
#define SRLHIP_F_CUR_TARGET     5   /* i32 current_target (2Target)          */
#define SRLHIP_F_LAST_REWARD    6   /* f64 reward of the last step, uncast   */
#define SRLHIP_F_EP_RETURN      7   /* f64 running episode return            */
#define SRLHIP_F_EP_LENGTH      8   /* i32 running episode length            */
#define SRLHIP_F_LAST_RETURN    11  /* f64 return of the last finished episode (Monitor's r) */
#define SRLHIP_F_LAST_LENGTH    12  /* i32 its length (Monitor's l)          */
#define SRLHIP_F_N_FINISHED     13  /* i32 episodes finished so far          */
#define SRLHIP_F_TARGET2_X      9   /* f64 second target (2Target)           */
#define SRLHIP_F_TARGET2_Y      10  /* f64 */
#define SRLHIP_F_KUKA_Q         16  /* f64[7]  arm joint positions           */
#define SRLHIP_F_KUKA_QD        17  /* f64[7]  arm joint velocities          */
#define SRLHIP_F_KUKA_EE_TARGET 18  /* f64[3]  Kuka.end_effector_pos         */
#define SRLHIP_F_KUKA_BUTTON_Q  19  /* f64[2]  glider position, velocity     */
#define SRLHIP_F_KUKA_BUTTON_POS 20 /* f64[3]  button_pos (target)           */
#define SRLHIP_F_KUKA_GRIPPER   21  /* f64[3]  getArmPos()                   */
#define SRLHIP_F_KUKA_COUNTERS  22  /* i32[3]  n_contacts, n_steps_outside, terminated */
#define SRLHIP_F_KUKA_BUTTON_XY 23  /* f64[2]  button base position on the table */
#define SRLHIP_F_KUKA_BUTTON2_Q 24  /* f64[2]  Kuka2Button: second glider position, velocity */
#define SRLHIP_F_KUKA_BUTTON2_XY 25 /* f64[2]  Kuka2Button: second button base position */
#define SRLHIP_F_KUKA_GOAL      26  /* i32[2]  Kuka2Button: goal_id, n_contacts[1] (n_contacts[0] is COUNTERS[0]) */
#define SRLHIP_F_KUKA_OBJECTS   27  /* f64[30] KukaRandButton: (x, y, present) of the ten distractor objects */
#define SRLHIP_F_KUKA_GRIPPER_Q 28  /* f64[5]  full model: joints 7, 8, 10, 11, 13 (gripper_to_arm, left finger, left tip, right finger, right tip) */
#define SRLHIP_F_KUKA_GRIPPER_QD 29 /* f64[5]  their velocities */
#define SRLHIP_F_KUKA_BODIES     30 /* f64[66] KukaRandButtonGymEnv, full model: (x y z vx vy vz) of the ten distractors (draw order) and the ball */
/* IK conditioning flag of the full model (no counterpart in the reference; it qualifies the parity claim of this library).
 * Kuka.applyAction's damped-least-squares IK (kuka.py:41-42 jd = 1e-5, kuka.py:144-156) has a gain of up to 1 / (2 sqrt(jd)) = 158
 * along a vanishing singular direction of the arm's Jacobian: while the arm is driven through the neighbourhood of a kinematic
 * singularity (e.g. the elbow joint through 0 with a saturating policy at the far edge of the workspace box), the closed loop
 * IK -> position motors -> IK amplifies float64 rounding differences by ~2.4x per step, so NO two float64 implementations (PyBullet
 * included) agree to 1e-4 on joint positions, or bit for bit on reward / done, after such a crossing.  Bit 0 of the value: sticky
 * per episode, set when an IK solve of the episode had det(J^T J + jd I) < SRLHIP_KUKA_IK_CROSS_DET (random-agent rollouts stay
 * above 1e-7, amplification starts near 1e-9), cleared by the episode's reset.  value >> 1: env-steps taken with the bit set since the
 * handle was created.  Parity with the oracle (tests/) is asserted on every env-step BEFORE the bit. */
#define SRLHIP_F_KUKA_IK_CROSSED 31 /* i32     full model: (flagged env-steps << 1) | sticky bit */
#define SRLHIP_KUKA_IK_CROSS_DET 3e-9
int srlhip_get_state(srlhip_handle h, int32_t field, void *out);
int srlhip_set_state(srlhip_handle h, int32_t field, const void *in);
/* Zero-copy hand-off of a field's device array (e.g. to torch via
 * __cuda_array_interface__). */
int srlhip_device_ptr(srlhip_handle h, int32_t field, void **dptr);

/* Replaces <Env>.render("rgb_array") (kuka_button_gym_env.py:370-420, mobile_robot_env.py:282-334)
 * for the whole batch: uint8 [num_envs][img_h][img_w][3 (6 with multi_view / fpv)] of the CURRENT state,
 * produced by the tile rasteriser.  Also what step()/reset()/rollout() write to obs_out when
 * cfg.obs_mode == SRLHIP_OBS_RAW_PIXELS.  img_out follows cfg.io_device. */
int srlhip_render(srlhip_handle h, void *img_out);

/* Per-env statistics of the most recently FINISHED episode and the number of
 * finished episodes: what stable_baselines.bench.Monitor records
 * (environments/utils.py:54).  HOST pointers, any may be NULL. */
int srlhip_episode_stats(srlhip_handle h, double *last_return, int32_t *last_length,
                         int32_t *n_finished);
/* The same two record planes WITHOUT a copy, for the per-step loop of rl_baselines.train (SubprocVecEnv's workers hand back
 * info['episode'] with the step's result, rl_baselines/utils.py:213-229; environments/utils.py:54): host-pointer handles
 * (cfg.io_device = 0) keep last_return [n] f64 / last_length [n] i32 in mapped pinned host memory that the step kernels write
 * directly.  The pointers stay valid until srlhip_destroy; entry i is final for the episode env i finished in a step once that
 * srlhip_step / srlhip_rollout call has returned.  EINVAL on device-pointer handles (use srlhip_episode_stats_device there). */
int srlhip_episode_records(srlhip_handle h, const double **last_return, const int32_t **last_length);

/* Same statistics for DEVICE-resident consumers (cfg.io_device-independent): enqueue-only on the handle's stream, no
 * host synchronisation.  last_return is narrowed to float32 — the element the multi-GPU path all-gathers over RCCL
 * (one all-gather of per-env episode returns per rollout).  DEVICE pointers, any may be NULL. */
int srlhip_episode_stats_device(srlhip_handle h, float *d_last_return, int32_t *d_last_length, int32_t *d_n_finished);

/* Stream ordering and live kernel timing (HIP events on the handle's stream). */
int srlhip_sync(srlhip_handle h);
/* Enqueue a copy of `bytes` bytes on the handle's stream, ordered with the steps / renders / encoder forwards enqueued there
 * (device <-> device, or device <-> PINNED host memory; the direction is taken from the pointers).  What a device-pointer caller
 * uses to feed actions and fetch reward / done / state planes without a second stream: the sharded HipVecEnv with a learned SRL
 * model moves only [n][state_dim] floats per step this way (the reference pickles a 150 KB frame per env and step through
 * MultiprocessSRLModel's queues, rl_baselines/utils.py:162-191).  Returns without waiting. */
int srlhip_copy_async(srlhip_handle h, void *dst, const void *src, size_t bytes);
int srlhip_stream(srlhip_handle h, void **hip_stream);
int srlhip_timing_begin(srlhip_handle h);
int srlhip_timing_end(srlhip_handle h, float *elapsed_ms);   /* synchronises */

/* HIP graphs for launch-bound step loops (cfg.io_device = 1 only): everything enqueued on the handle's stream between
 * graph_begin and graph_end — srlhip_step / srlhip_rollout / srlhip_render with device pointers, srlhip_encoder_forward
 * on srlhip_stream() — is captured instead of executed; srlhip_graph_launch() replays it (same buffers) with one launch.
 * Make one eager call of the same sequence first: lazily allocated scratch must exist before the capture. */
typedef struct srlhip_graph *srlhip_graph_handle;
int srlhip_graph_begin(srlhip_handle h);
int srlhip_graph_end(srlhip_handle h, srlhip_graph_handle *out);
int srlhip_graph_launch(srlhip_handle h, srlhip_graph_handle g);
int srlhip_graph_destroy(srlhip_graph_handle g);

/* The part of the Kuka-button model that is NOT pinned by the reference's own source: link frames, inertial parameters,
 * joint limits / damping of pybullet_data/kuka_iiwa/kuka_with_gripper2.sdf (loaded at kuka.py:60), the IK / gripper
 * reference points, the gripper's collision spheres, the table and button-base heights of table/table.urdf — recalled
 * from upstream in this repo, so kept as DATA: 138 doubles.  srlhip_kuka_default_model() returns the baked
 * table; srlhip_set_kuka_model() installs another one on a Kuka handle (before the next srlhip_reset): the settled state is
 * re-integrated, every following reset / step uses it.  A 7-joint serial arm whose joints turn about their local z is
 * assumed (joint_rpy = URDF rpy of the joint frame in the parent link frame, joint_xyz its origin; inertia = principal
 * moments about com, axes parallel to the link frame).  With a table installed the batch is stepped by the lane-group
 * kernel at any size; lumped Kuka2ButtonGymEnv handles return SRLHIP_ENOTSUP (full-model ones take srlhip_set_kuka_tree_model).  tests/golden/make_kuka_pybullet_golden.py
 * fills this struct from pybullet_data when PyBullet is importable. */
typedef struct srlhip_kuka_model {
    double joint_xyz[7][3], joint_rpy[7][3], joint_lower[7], joint_upper[7], joint_damping;
    double mass[7], com[7][3], inertia[7][3];
    double ee_point[3];          /* IK end effector (link_7 inertial frame) in the link_7 frame          */
    double gripper_point[3];     /* COM of gripper link 8 (getArmPos) in the link_7 frame                 */
    double sphere[6][4];         /* gripper collision spheres: centre in the link_7 frame, radius         */
    double table_top_z, button_base_z;
} srlhip_kuka_model;
int srlhip_kuka_default_model(srlhip_kuka_model *m);
int srlhip_set_kuka_model(srlhip_handle h, const srlhip_kuka_model *m);

/* The full model as data (SRLHIP_KUKA_MODEL_FULL): a kinematic tree of up to 12 revolute DoFs, parents before children.
 * Integer-valued fields are stored as doubles so that the struct is a flat table of 510 doubles.  Per DoF: parent (-1 = the fixed
 * base at kuka.py:63's pose), joint frame in the parent link (origin, fixed rotation Rj row-major: child = Rj * Rot(axis, q)),
 * unit axis in the joint frame, limits (lower > upper = none), damping, link mass / centre of mass / inertia about it in link axes
 * (xx xy xz yy yz zz; fixed-joint children are merged in exactly), the joint's POSITION_CONTROL motor (positionGain, force,
 * maxVelocity; velocityGain is 1), the pybullet joint index.  Then the IK end-effector link / point (kuka_end_effector_index 6),
 * the getArmPos() link / point (kuka_gripper_index 8: its COM), up to 16 collision spheres (link, centre, radius, combined lateral
 * friction), table / button-base heights, the per-step budget of limit + contact-normal rows (<= 6: the lane group's second row bank; values above are rejected) and the
 * friction switch, then the SOLVER DETAILS below.
 * Impulse bounds: a contact-normal row is boxed to [0, 1e10] (Bullet's default upper bound) in the oracle and in the kernel's general
 * path; the one-button contact fast path solves normal rows in u = lambda / 2^33 so that the projection is the hardware clamp of one add,
 * i.e. its upper bound is 2^33 = 8.59e9 — impulses are ~1e-2, neither bound is ever reached, the results are identical.
 * The arm part is in-tree or pinned elsewhere (srlhip_kuka_model); the gripper part is RECALLED from kuka_with_gripper2.sdf
 * [UNVERIFIED-MEMORY] — tests/golden/make_kuka_pybullet_golden.py overwrites it from pybullet_data when PyBullet is importable. */
typedef struct srlhip_kuka_tree_joint {
    double parent, xyz[3], Rj[9], axis[3], lower, upper, damping, mass, com[3], inertia[6], kp, max_force, max_vel, joint_index;
} srlhip_kuka_tree_joint;
typedef struct srlhip_kuka_tree_sphere { double link, c[3], r, mu; } srlhip_kuka_tree_sphere;
typedef struct srlhip_kuka_tree_model {
    double nd;
    srlhip_kuka_tree_joint j[12];
    double ee_link, ee_point[3], grip_link, grip_point[3], nsphere;
    srlhip_kuka_tree_sphere s[16];
    double table_top_z, button_base_z, max_generic_rows, friction;
    /* Details of the dependency's constraint solver (Bullet 2.87 btMultiBodyConstraintSolver as driven by pybullet 1.8.6:
     * kuka_button_gym_env.py:219-220 sets 150 iterations, :351 steps it) that this repo RECALLS and cannot read — kept as data so
     * that the day a PyBullet fixture exists, matching it is a choice of table values (tests/golden/fit_kuka_pin.py searches them on
     * the oracle), not a kernel change.  Defaults (0, 0.2, 0.2, 0) = the behaviour of rounds 1-3.
     *   solver_detail   bit mask of SRLHIP_KUKA_DETAIL_*
     *   contact_erp     error reduction of penetrating contact rows (Bullet m_erp2; pybullet's server is recalled to use 0.08)
     *   limit_erp       the same for joint-limit rows, the button's two stops included (Bullet m_erp)
     *   linear_slop     a contact row sees penetration = distance + linear_slop (Bullet m_linearSlop; recalled server value 1e-5) */
    double solver_detail, contact_erp, limit_erp, linear_slop;
} srlhip_kuka_tree_model;
#define SRLHIP_KUKA_DETAIL_ALT_SWEEP  1  /* the non-contact rows (motors, limits, button rows) are swept backwards on even iterations */
#define SRLHIP_KUKA_DETAIL_BODY_ORDER 2  /* non-contact rows in body-creation order: button (stops, motor) before the arm (limits, motors) */
#define SRLHIP_KUKA_DETAIL_FRICTION2  4  /* second friction row per contact along n x t1; the row budget becomes min(max_generic_rows, 4) */
int srlhip_kuka_tree_default_model(srlhip_kuka_tree_model *m);                   /* host only, no GPU needed */
/* Install a table on a SRLHIP_KUKA_MODEL_FULL handle: settled state and start-state table are re-integrated; reset afterwards. */
int srlhip_set_kuka_tree_model(srlhip_handle h, const srlhip_kuka_tree_model *m);

/* Which kernel steps this Kuka handle's batch: 2 = tree lane-group (full model, kuka_tree_rollout_k: every batch size), 1 = lane-group (16 lanes per env, kuka_group_rollout_k: batches up to 12288
 * envs), 0 = lane-per-env (kuka_rollout_k: larger batches and the lumped Kuka2ButtonGymEnv); the environment variable
 * SRLHIP_KUKA_KERNEL=group|lane overrides the choice.  Both read and write the same state and produce the same outputs
 * (to ~1e-11 on joint positions; discrete flags identical).  The tree kernel has a two-wavefronts-per-SIMD variant
 * (kuka_tree_rollout_occ_k: one-button envs, Cartesian actions) that the library uses from 65536 envs up;
 * SRLHIP_KUKA_OCC=0|1 forces either.  Same state, same outputs to the same bar. */
int srlhip_kuka_kernel(srlhip_handle h);

/* Diagnostic: runs every cross-lane primitive of the lane-group Kuka kernel (csrc/kuka_group.hpp: DPP row broadcasts and
 * shifts, row votes, the fused solver-row instructions, the lane-parallel Gauss-Jordan, the prefix-composed forward
 * kinematics for joint angles q7) on one wavefront of device_id and returns their per-lane results, out[40][64] doubles.
 * tests/test_gpu_group_primitives.py checks them against the definitions the CPU-side emulation of the same source uses. */
int srlhip_selftest_group_primitives(int32_t device_id, const double *q7, double *out, int32_t out_doubles);

const char *srlhip_last_error(srlhip_handle h);
int srlhip_abi_version(void);

/* ---- SRL encoder (raw_pixels -> state) ----------------------------------------------------------------
 * Replaces SRLNeuralNetwork.getState (state_representation/models.py:178-193: preprocessImage, the H/W
 * transpose, one CustomCNN forward) and the one-image-at-a-time encoder server MultiprocessSRLModel._run
 * (rl_baselines/utils.py:181-191) for a whole DEVICE-resident batch of frames.  Weights are plain float32 host
 * arrays in torch layout with the BatchNorms already folded into the preceding convolution
 * (w' = w*gamma/sigma, b' = beta - mu*gamma/sigma):
 *   conv1_w [64][C][7][7] conv1_b [64]   conv2_w, conv3_w [64][64][3][3]   conv2_b, conv3_b [64]
 *   fc_w [state_dim][64 * cells]  fc_b [state_dim]      (cells = the last pooled map's size, torch flatten order)
 * Covered shapes (srlhip_encoder_supported() tells beforehand, anything else -> SRLHIP_ENOTSUP and the caller keeps
 * the PyTorch-ROCm forward): frames of 8..1024 pixels a side that survive the three conv + pool stages, with C = 3
 * channels or C = 6 (multi_view / fpv: two cameras, kuka_button_gym_env.py:401-417).  64x64x3 (BASELINE configs 4-5)
 * runs as ONE fused kernel (csrc/encoder.hip); every other shape — the reference's 224x224 RENDER size among them —
 * runs layer by layer with f16 hi/lo activation planes in HBM (csrc/encoder_general.hip), same arithmetic.
 * images_dev uint8 [n][H][W][C] and states_dev float [n][state_dim] are DEVICE pointers on device_id; the call only
 * enqueues on hip_stream (NULL = the default stream).  The layered path keeps grow-only scratch planes per handle: a
 * caller that captures the forward into a HIP graph calls it once with its batch size before capturing. */
typedef struct srlhip_encoder *srlhip_encoder_handle;
int srlhip_encoder_supported(int32_t img_h, int32_t img_w, int32_t n_channels);
/* Host-only: inputs of the fully connected layer for this frame shape (64 channels x the last pooled map), 0 when the
 * shape is not covered — the second dimension fc_w must have. */
int32_t srlhip_encoder_feature_count(int32_t img_h, int32_t img_w, int32_t n_channels);
int srlhip_encoder_create(int32_t device_id, int32_t img_h, int32_t img_w, int32_t n_channels, int32_t state_dim,
                          const float *conv1_w, const float *conv1_b, const float *conv2_w, const float *conv2_b,
                          const float *conv3_w, const float *conv3_b, const float *fc_w, const float *fc_b,
                          srlhip_encoder_handle *out);
int srlhip_encoder_forward(srlhip_encoder_handle e, const uint8_t *images_dev, int32_t n, float *states_dev,
                           void *hip_stream);
/* Synchronises the device; *flag = 1 if an activation ever left float16's range (|x| >= 65504: the split-f16
 * matrix path then lost accuracy and the caller should use the PyTorch forward for these weights). */
int srlhip_encoder_overflow(srlhip_encoder_handle e, int32_t *flag);
/* Diagnostic: one synchronous forward whose workgroup 0 stamps the shader clock (s_memtime) at nine points of its
 * first frames; cycles9[k] = cycles between stamp k and k+1, slowest wave, averaged over frames:
 * 0 unpack, 1 layer-1 MFMA + pooling, 2 barrier, 3 layer-2 k-loop, 4 layer-2 epilogue, 5 barrier, 6 layer 3,
 * 7 FC, 8 loop turn-around.  Used by bench.py / DESIGN.md to attribute the kernel's time. */
int srlhip_encoder_phase_cycles(srlhip_encoder_handle e, const uint8_t *images_dev, int32_t n, float *states_dev,
                                int64_t *cycles9);
int srlhip_encoder_destroy(srlhip_encoder_handle e);
const char *srlhip_encoder_last_error(srlhip_encoder_handle e);
/* Host-only: the MFMA B-operand image create() uploads (normalisation folded into layer 1, every weight split
 * into f16 hi/lo), srlhip_encoder_pack_bytes() bytes.  Exposed so the packing can be checked without a GPU. */
/* Host-only, layered path: layer 1's B-operand image for 3- or 6-channel frames — [channel half][k-step][lane][8 hi | 8 lo]
 * f16 with k = (ky * 8 + kx) * cpix + c (cpix = 4 or 8 staged f16 per pixel: the channels, the validity mask, zero fill; kx = 7
 * is a zero slot), 2 * (7 * 8 * cpix / 16) * 2048 bytes; *scale = the power-of-two pre-scale. */
int srlhip_encoder_pack_first_layer(int32_t n_channels, const float *conv1_w, const float *conv1_b, void *out, size_t out_bytes, float *scale);
size_t srlhip_encoder_pack_bytes(void);
/* Host-only: layer 1 of the fused 64x64x3 kernel as it runs since round 6 — on the INT8 matrix pipe, exactly.  out = [channel half]
 * [kernel row 0..6][digit 0..2][lane][16 i8] (srlhip_encoder_pack_i8_bytes() bytes): the folded weight of output channel o at k =
 * row * 32 + slot * 4 + c (zero_slot = 0, the fused kernel: slot 0 empty, slot s = kernel column s - 1; zero_slot = 7, the layered
 * kernels of 3-channel frames of any other size: slot s = kernel column s, slot 7 empty; c < 3 the colour taps w / (255 std_c) that multiply p - 128,
 * c = 3 the validity-mask tap [sum_c w_c (128 / 255 - mean_c) / std_c plus the folded bias on the centre tap] / 127: the mask
 * byte of an inside pixel is 127) is the 24-bit
 * fixed-point number (65536 d0 + 256 d1 + d2) / scale_o with balanced digits d in [-128, 127]; inv_scale64[o] = 256 / scale_o, a
 * power of two.  Lane (h, n) of a fragment holds k = 16 h .. 16 h + 15 of output channel 32 * half + n. */
size_t srlhip_encoder_pack_i8_bytes(void);
int srlhip_encoder_pack_i8(const float *conv1_w, const float *conv1_b, int32_t zero_slot, void *out, size_t out_bytes, float *inv_scale64);
int srlhip_encoder_pack(const float *conv1_w, const float *conv1_b, const float *conv2_w, const float *conv3_w,
                        void *out, size_t out_bytes, float *scales3 /* power-of-two pre-scale of layers 1..3 */);


#ifdef __cplusplus
}
#endif
#endif /* SRLHIP_H */
