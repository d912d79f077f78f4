// kuka_env.hpp — env-level logic of KukaButtonGymEnv on top of kuka_core.hpp:
//   reset   kuka_button_gym_env.py:214-281   (settled state + 5 random init actions)
//   step    :293-340, step2 :342-368, _termination :422-426, _reward :428-463, getSRLState :175-189
// and of its variants: KukaMovingButtonGymEnv (runtime flag cfg.moving) and Kuka2ButtonGymEnv
// (kuka_2button_gym_env.py:33-200; compile-time NB == 2: a second button body, goal switching, its own reward).
//
// Reset is O(1) on the device.  The 500 settle steps are RNG- and contact-free,
// and each of the 5 init actions is one of only six (sign, axis) moves of 0.03 m
// applied without noise, so an episode can only start from one of 6^5 = 7776
// arm states (32 for continuous actions).  Those are integrated once per handle
// with the same physics_step() and kept as a table in HBM; reset draws the RNG
// exactly like the reference, turns the draws into a table index and gathers
// 288 bytes.  (action_joints mode has a continuous init distribution and
// integrates its 5 steps in the kernel.)
#pragma once
#include "kuka_core.hpp"

namespace srl {
namespace kuka {

constexpr int kStartDoubles = 36;          // q7 qd7 sq7 cq7 ee3 bq bqd grip3
constexpr int kNumStartsDiscrete = 7776;   // 6^5
constexpr int kNumStartsContinuous = 32;   // 2^5

SRL_HD void pack_start(const Env &e, double *o) {
#pragma unroll
    for (int i = 0; i < ND; i++) { o[i] = e.q[i]; o[7 + i] = e.qd[i]; o[14 + i] = e.sq[i]; o[21 + i] = e.cq[i]; }
#pragma unroll
    for (int k = 0; k < 3; k++) { o[28 + k] = e.ee[k]; o[33 + k] = e.grip[k]; }
    o[31] = e.bq; o[32] = e.bqd;
}
SRL_HD void unpack_start(Env &e, const double *o) {
#pragma unroll
    for (int i = 0; i < ND; i++) { e.q[i] = o[i]; e.qd[i] = o[7 + i]; e.sq[i] = o[14 + i]; e.cq[i] = o[21 + i]; }
#pragma unroll
    for (int k = 0; k < 3; k++) { e.ee[k] = o[28 + k]; e.grip[k] = o[33 + k]; }
    e.bq = o[31]; e.bqd = o[32];
}

// state right after loadSDF/resetJointState (kuka.py:56-73), before the settle steps
SRL_HD void initial_env(Env &e) {
#pragma unroll
    for (int i = 0; i < ND; i++) { e.q[i] = kJointPositions[i]; e.qd[i] = 0.0; }
#pragma unroll
    for (int k = 0; k < 3; k++) { e.ee[k] = kEeInit[k]; e.bpos[k] = 0.0; }
    e.bq = 0.0; e.bqd = 0.0; e.bx = kButtonX; e.by = kButtonY; e.bz = kButtonBaseZ; e.bspeed = 0.0;
    e.motor_on = 0; e.contact_button = 0; e.contact_table = 0;
    e.counter = 0; e.n_contacts = 0; e.n_outside = 0; e.terminated = 0;
    e.b2q = 0.0; e.b2qd = 0.0; e.b2x = kButtonX; e.b2y = kButton2Y2B;
    e.contact_body1 = 0; e.contact_body2 = 0; e.goal_id = 0; e.n_contacts2 = 0;
    update_trig_and_gripper(e);
}

// one init action of reset(): code = sign_bit * 3 + axis (discrete) or sign_bit (continuous) -> Cartesian increment
SRL_HD void init_action_motor(const Cfg &cfg, int code, double motor[3]) {
    const int axis = code % 3;
    const double sign = code >= 3 ? 1.0 : -1.0, dir = code ? 1.0 : -1.0;
#pragma unroll
    for (int k = 0; k < 3; k++) motor[k] = cfg.is_discrete ? (k == axis ? sign * kDeltaV : 0.0) : kDeltaVContinuous * dir;
}

SRL_HD double norm3(const double a[3], const double b[3]) {      // np.linalg.norm(a - b, 2): ddot with fma
    const double d0 = a[0] - b[0], d1 = a[1] - b[1], d2 = a[2] - b[2];
    return sqrt(fma(d2, d2, fma(d1, d1, fma(d0, d0, 0.0))));
}
SRL_HD bool termination(const Env &e, const Cfg &cfg) { return e.terminated || e.counter > cfg.max_steps; }

// Kuka2ButtonGymEnv._reward (kuka_2button_gym_env.py:141-200).  bpos = button_all_pos[goal_id].
SRL_HD double reward_two(Env &e, const Cfg &cfg) {
    const double distance = norm3(e.bpos, e.grip);
    const int contact = e.goal_id ? e.contact_body2 : e.contact_body1;
    int reward = 0;
    if (e.goal_id) { e.n_contacts2 += contact; reward = contact; }      // sparse reward only on the last button
    else e.n_contacts += contact;
    if (e.goal_id == 0 && e.n_contacts >= kNContactsBeforeTermination) {  // first button pressed: switch the goal
        e.goal_id = 1;
        e.bpos[0] = e.b2x; e.bpos[1] = e.b2y;                             // z is Z_TABLE + 0.28 for both
    }
    if (distance > cfg.max_distance || e.contact_table) { reward = -1; e.n_outside += 1; }
    else e.n_outside = 0;
    if (e.contact_table || e.n_contacts2 >= kNContactsBeforeTermination || e.n_outside >= kNStepsOutside - 1) e.terminated = 1;
    if (cfg.shape_reward) {
        const int nc_goal = e.goal_id ? e.n_contacts2 : e.n_contacts;
        if (e.terminated && reward > 0) return 50.0;
        if (nc_goal < kNContactsBeforeTermination && contact) return 25.0;
        if (e.contact_table) return -250.0;
        if (distance > cfg.max_distance) return -20.0;
        return -distance;
    }
    return (double)reward;
}

SRL_HD double reward_fn(Env &e, const Cfg &cfg) {
    const double distance = norm3(e.bpos, e.grip);
    int reward = e.contact_button ? 1 : 0;
    e.n_contacts += reward;
    if (distance > cfg.max_distance || e.contact_table) { reward = -1; e.n_outside += 1; }
    else e.n_outside = 0;
    if (e.contact_table || e.n_contacts >= kNContactsBeforeTermination || e.n_outside >= kNStepsOutside) e.terminated = 1;
    if (cfg.shape_reward) {
        if (cfg.is_discrete && !cfg.moving) return -distance;     // MovingButton: 50 / -250 / -d for every action type
        if (e.terminated && reward > 0) return 50.0;
        if (e.terminated && reward < 0) return -250.0;
        return -distance;
    }
    return (double)reward;
}

// RNG adaptor over caller-supplied draws (SRLHIP_RNG_HOST)
struct HostDraws {
    const double *v; int i;
    SRL_HD double double01() { return v[i++]; }
    SRL_HD double uniform(double, double) { return v[i++]; }
    SRL_HD double normal(double, double) { return v[i++]; }
    SRL_HD uint32_t bounded(uint32_t) { return (uint32_t)v[i++]; }
};

// Everything KukaButtonGymEnv.reset draws from the env's RNG, in the reference's order (kuka_button_gym_env.py:214-255 and
// the variants' overrides), turned into what the state assembly needs.  Shared by the lane-per-env and the lane-group kernels.
struct ResetDraw {
    double bx, by, speed, b2x, b2y;
    int idx;                        // row of the start-state table (Cartesian action modes)
    double g[kNInitActions];        // joints mode: 7 + N(0,1) of each init action
};
struct NoObjectHook { SRL_HD void operator()(int, double, double, bool) const {} };
// `hook(i, ox, oy, keep)` sees every distractor candidate of KukaRandButtonGymEnv as it is drawn (the tree kernel's free bodies)
template <int NB, class R, class H = NoObjectHook>
SRL_HD void reset_draw(const Cfg &cfg, double *objs, int64_t objs_stride, R &rng, ResetDraw &d, H hook = H()) {
#pragma clang fp contract(off)   // wrapper arithmetic is numpy's: unfused (physics_step keeps the file's setting)
    d.bx = kButtonX; d.by = kButtonY; d.speed = 0.0; d.b2x = kButtonX; d.b2y = kButton2Y2B; d.idx = 0;
    if (cfg.moving) d.speed = 0.001 * (rng.bounded(1) ? 1.0 : -1.0);   // BUTTON_SPEED * np_random.choice([-1, 1]), drawn first
    if constexpr (NB == 2) {                                          // kuka_2button_gym_env.py:55-70
        if (cfg.random_target) { (void)rng.uniform(-1, 1); (void)rng.uniform(0, 1); }   // overwritten two lines later
        d.bx = 0.5 + 0.0 * rng.uniform(-1, 1); d.by = kButton1Y2B + 0.0 * rng.uniform(-1, 1);
        if (cfg.random_target) { d.b2x += 0.15 * rng.uniform(-1, 1); d.b2y += 0.175 * rng.uniform(-1, 0); }
    } else if (cfg.random_target) { d.bx += 0.15 * rng.uniform(-1, 1); d.by += 0.3 * rng.uniform(-1, 1); }
    if (cfg.rand_objects) {                                           // kuka_rand_button_gym_env.py:62-69: two draws per candidate
        for (int i = 0; i < 10; i++) {
            const double ox = 0.5 + 0.15 * rng.uniform(-1, 1), oy = 0 + 0.3 * rng.uniform(-1, 1);
            const bool keep = (ox < d.bx - 0.1) || (ox > d.bx + 0.1) || (oy < d.by - 0.1) || (oy > d.by + 0.1);
            if (objs) { objs[(3 * i) * objs_stride] = ox; objs[(3 * i + 1) * objs_stride] = oy; objs[(3 * i + 2) * objs_stride] = keep ? 1.0 : 0.0; }
            hook(i, ox, oy, keep);
        }
    }
    if (!cfg.is_discrete && cfg.action_joints) {
        // np_random.normal(joints.shape): the shape tuple is `loc` -> one draw 7 + N(0,1) per init action,
        // broadcast over the joints.  All five are drawn before the physics steps.
        for (int k = 0; k < kNInitActions; k++) d.g[k] = rng.normal(7.0, 1.0);
    } else {
        int mul = 1;
        for (int k = 0; k < kNInitActions; k++) {
            int code;
            if (cfg.is_discrete) {
                const int sign_bit = rng.double01() > 0.5 ? 1 : 0;
                const int axis = (int)rng.bounded(2);
                code = sign_bit * 3 + axis;
                d.idx += code * mul; mul *= 6;
            } else {
                // np_random.normal((3,)) -> one draw 3 + N(0,1); L2-normalised 1-vector = +-1, broadcast
                const double g = rng.normal(3.0, 1.0);
                code = (g / sqrt(fma(g, g, 0.0))) > 0 ? 1 : 0;
                d.idx += code * mul; mul *= 2;
            }
        }
    }
}
// scalar part of the state after reset (everything but the arm / glider state, which comes from the start table or from
// the five joint-mode init steps)
template <int NB>
SRL_HD void reset_finish(Env &e, const ResetDraw &d, double base_z = kButtonBaseZ) {
#pragma clang fp contract(off)
    e.bx = d.bx; e.by = d.by;
    if constexpr (NB == 2) { e.b2x = d.b2x; e.b2y = d.b2y; }
    e.bz = base_z; e.bspeed = d.speed;
    e.bpos[0] = d.bx; e.bpos[1] = d.by;
    e.bpos[2] = base_z + kGliderOriginZ + e.bq + kButtonDistanceHeight;
    if constexpr (NB == 2) { e.bpos[2] = kZTable + kButtonDistanceHeight; e.goal_id = 0; e.n_contacts2 = 0; e.contact_body1 = 0; e.contact_body2 = 0; }
    e.counter = 0; e.n_contacts = 0; e.n_outside = 0; e.terminated = 0; e.ikx &= ~1;
}

// KukaButtonGymEnv.reset.  `starts` = table of episode start states, `settled` = state after the settle steps.
template <int NB, class R>
SRL_HD void reset_env(Env &e, const Cfg &cfg, const Scratch &sc, R &rng, const double *starts, const double *settled) {
#pragma clang fp contract(off)   // wrapper arithmetic is numpy's: unfused (physics_step keeps the file's setting)
    ResetDraw d;
    reset_draw<NB>(cfg, sc.objs, sc.gst, rng, d);
    e.motor_on = 0; e.contact_button = 0; e.contact_table = 0;
    if (!cfg.is_discrete && cfg.action_joints) {
        unpack_start(e, settled);
        e.bx = d.bx; e.by = d.by; e.bz = kButtonBaseZ;
        if constexpr (NB == 2) { e.b2x = d.b2x; e.b2y = d.b2y; e.b2q = e.bq; e.b2qd = e.bqd; }
        // (arrays handed to physics_step live outside the loop: per-iteration locals next to the inlined step have been
        //  seen to be mis-coloured onto live state by the gfx950 backend, DESIGN.md §Kuka kernel, compiler notes)
        double joints[ND], motor[3] = {0, 0, 0};
        for (int k = 0; k < kNInitActions; k++) {
#pragma unroll
            for (int j = 0; j < ND; j++) joints[j] = kJointPositions[j] + kDeltaTheta * d.g[k];
            physics_step<NB>(e, cfg, sc, motor, true, joints);
        }
    } else {
        unpack_start(e, starts + (int64_t)d.idx * kStartDoubles);
        e.bx = d.bx; e.by = d.by;
        if constexpr (NB == 2) { e.b2x = d.b2x; e.b2y = d.b2y; e.b2q = e.bq; e.b2qd = e.bqd; }   // same urdf, same free steps
    }
    reset_finish<NB>(e, d);
}

SRL_HD void observe(const Env &e, const Cfg &cfg, float *o, int64_t stride) {   // getSRLState
    int k = 0;
    if (cfg.obs_mode == 0 || cfg.obs_mode == 2)
        for (int j = 0; j < 3; j++) o[(k++) * stride] = (float)(e.grip[j] - e.bpos[j]);
    if (cfg.obs_mode == 1 || cfg.obs_mode == 2)
        for (int j = 0; j < 14; j++) o[(k++) * stride] = (float)kJointPositions[j];
}

// First half of KukaButtonGymEnv.step (:293-340): the moving-button update, the noise draw and the action mapping.
// Cartesian modes fill motor[3]; joints mode returns the noisy scale `dth` (targets = a_j * dth + q0_j, built by the
// caller); action < 0 == None (no RNG draw, :295-299).  Shared by the lane-per-env and the lane-group kernels.
struct StepCmd { double motor[3]; double dth; bool joint_mode; bool scaled_joints; };
template <class R>
SRL_HD void step_command(Env &e, const Cfg &cfg, R &rng, int action, const float *ca3, StepCmd &c) {
#pragma clang fp contract(off)
    c.motor[0] = 0; c.motor[1] = 0; c.motor[2] = 0; c.dth = 0; c.joint_mode = false; c.scaled_joints = false;
    if (cfg.moving) {                                            // kuka_moving_button_gym_env.py:111-119
        if (e.bpos[1] > 0.3 || e.bpos[1] < -0.3) e.bspeed = -e.bspeed;
        e.bpos[1] += e.bspeed;
        e.by = e.bpos[1];
        e.bz = e.bpos[2] - kButtonDistanceHeight;                // base re-placed at the recorded cap height
    }
    if (action < 0) {
        c.joint_mode = cfg.action_joints != 0;
    } else if (cfg.is_discrete) {
        const double dv = kDeltaV + rng.normal(0.0, kNoiseStd);
        if (action == 0) c.motor[0] = -dv; else if (action == 1) c.motor[0] = dv;
        else if (action == 2) c.motor[1] = -dv; else if (action == 3) c.motor[1] = dv;
        else if (action == 4) c.motor[2] = -dv; else if (action == 5) c.motor[2] = cfg.force_down ? -dv : dv;
    } else if (cfg.action_joints) {
        c.dth = kDeltaTheta + rng.normal(0.0, kNoiseStdJoints);
        c.joint_mode = true; c.scaled_joints = true;
    } else {
        const double dv = kDeltaVContinuous + rng.normal(0.0, kNoiseStdContinuous);
        c.motor[0] = (double)ca3[0] * dv; c.motor[1] = (double)ca3[1] * dv;
        c.motor[2] = cfg.force_down ? -fabs((double)ca3[2] * dv) : (double)ca3[2] * dv;
    }
    e.motor_on = 1;
}
// joint target of joint j in joints mode: float32 action * python float stays float32, + float64 list -> float64
SRL_HD double joint_target(const StepCmd &c, float a, double q0) {
#pragma clang fp contract(off)
    return c.scaled_joints ? (double)(a * (float)c.dth) + q0 : q0;
}

// KukaButtonGymEnv.step + step2.  action < 0 == None.  Returns the reward; *done = _termination().
template <int NB, class R>
SRL_HD double env_step(Env &e, const Cfg &cfg, const Scratch &sc, R &rng, int action, const float *ca, bool *done) {
    StepCmd c;
    step_command(e, cfg, rng, action, ca, c);
    double joints[ND];
#pragma unroll
    for (int j = 0; j < ND; j++) joints[j] = joint_target(c, ca[j], kJointPositions[j]);
    for (int rep = 0; rep < cfg.action_repeat; rep++) {
        physics_step<NB>(e, cfg, sc, c.motor, c.joint_mode, joints);
        if (termination(e, cfg)) break;
        e.counter += 1;
    }
    const double reward = NB == 2 ? reward_two(e, cfg) : reward_fn(e, cfg);
    *done = termination(e, cfg);
    return reward;
}

}  // namespace kuka
}  // namespace srl
