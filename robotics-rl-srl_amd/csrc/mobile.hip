// mobile.hip — MobileRobot family stepper for gfx950 (MI355X).
//
// Replaces, for a whole batch of envs per launch:
//   reset   mobile_robot_env.py:159-222 (+ variants: 1D :58-74, 2Target :35-67,
//           LineTarget :56-64) — only the RNG draws and state init; every
//           pybullet call on this path is scene loading / rendering.
//   step    mobile_robot_env.py:235-280, _reward :345-363, _termination :336-343
//           (2Target reward :162-181, LineTarget reward :108-125, 1D step :108-147)
//
// Mapping: one lane per env, structure-of-arrays state (env index fastest) so
// a wavefront's 64 loads/stores of a field are one coalesced 512-byte row.
// Arithmetic is float64 with -ffp-contract=off, the reference's numpy f64; the
// only fused op is the explicit fma chain of np.linalg.norm's ddot (see
// step_env()).  ~40 flops per env-step: the path is HBM/launch bound, no LDS, no
// MFMA.  mobile_rollout_k keeps the state in VGPRs for T steps and streams the
// [T][N] observation / reward / done planes with non-temporal stores.
//
// The sequential part of a rollout is only the state recurrence.  The synthetic agent's actions do not depend
// on the state, so they are drawn by a separate, fully parallel kernel (mobile_sample_actions_k: one thread per
// (step, env), Philox is random access) into the [T][N] action plane; the recurrence kernel then reads them one
// step ahead of use (software-pipelined load) and is specialised at compile time on the env kind and the action type.
// At the 4096-env BASELINE size only 64 wavefronts exist, so the launch time is T x (cycles per step of one
// wavefront): taking the 10-round Philox block out of that chain is what matters — and so is cutting the chain
// itself: with counter-based streams the episodes of an env are independent of each other, so a T-step rollout runs
// as ceil(T / 251) + 1 segments per env on separate lanes (mobile_rollout_ep_k below).
// Round 4: what is left is the issue stream of ONE wavefront over 251 steps, so the step was cut from ~200 to 57 instructions —
// the shaped reward's float64 square root is a template parameter (with a run-time flag it was evaluated every step and
// selected away), interior steps run in straight-line chunks of 8 (no per-step predicate, no reset body to branch around, no
// plane checks; the next chunk's actions are requested ahead of the chunk's stores), and the sampler workgroups index the plane
// without 64-bit divisions: 0.095 -> 0.053 ms per 4096-env x 2048-step rollout with the synthetic agent (of which ~15 us are the
// sampler workgroups' 8.4 M Philox blocks sharing the SIMDs), 0.0375 ms with caller-supplied actions
// (profiles/probes/mobile_chain_probe.py, profiles/r04_mobile_*).
#include "internal.hpp"

namespace srl {

namespace {
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kBlock = 256;

// np.linalg.norm(v, 2) == sqrt(ddot(v, v)); OpenBLAS accumulates with FMA: see step_env.

struct MobileEnv {
    double x, y, tx, ty, t2x, t2y;
    int32_t counter, cur;
};

__device__ __forceinline__ void load_env(const MobileState &s, int e, MobileEnv &m) {
    m.x = s.pos_x[e]; m.y = s.pos_y[e]; m.tx = s.tgt_x[e]; m.ty = s.tgt_y[e];
    m.t2x = s.tgt2_x[e]; m.t2y = s.tgt2_y[e]; m.counter = s.counter[e]; m.cur = s.cur_target[e];
}
__device__ __forceinline__ void store_env(const MobileState &s, int e, const MobileEnv &m) {
    s.pos_x[e] = m.x; s.pos_y[e] = m.y; s.tgt_x[e] = m.tx; s.tgt_y[e] = m.ty;
    s.tgt2_x[e] = m.t2x; s.tgt2_y[e] = m.t2y; s.counter[e] = m.counter; s.cur_target[e] = m.cur;
}

// Random sources ---------------------------------------------------------------
struct HostRng {                 // SRLHIP_RNG_HOST: caller pre-drew everything
    const double *rand; int i; double noise;
    __device__ double uniform(double, double) { return rand[i++]; }
    __device__ double normal(double, double) { return noise; }
};
struct PhiloxRng {               // SRLHIP_RNG_PHILOX
    Philox p;
    __device__ double uniform(double lo, double hi) { return p.uniform(lo, hi); }
    // scale == 0 draws nothing: counter streams need no alignment with numpy
    __device__ double normal(double loc, double scale) { return scale == 0.0 ? loc : p.normal(loc, scale); }
};
struct MtRng {                   // SRLHIP_RNG_MT19937: np_random itself
    Mt19937 m;
    __device__ double uniform(double lo, double hi) { return m.uniform(lo, hi); }
    __device__ double normal(double loc, double scale) { return m.normal(loc, scale); }
};

template <int MODE> struct RngSel;
template <> struct RngSel<SRLHIP_RNG_HOST> { using type = HostRng; };
template <> struct RngSel<SRLHIP_RNG_PHILOX> { using type = PhiloxRng; };
template <> struct RngSel<SRLHIP_RNG_MT19937> { using type = MtRng; };

template <int MODE>
__device__ __forceinline__ void rng_load(typename RngSel<MODE>::type &r, const RngState &rs, int e, int n,
                                         const double *host_rand, int rand_stride, const double *noise) {
    if constexpr (MODE == SRLHIP_RNG_HOST) {
        r.rand = host_rand ? host_rand + (int64_t)e * rand_stride : nullptr;
        r.i = 0;
        r.noise = noise ? noise[e] : 0.0;
    } else if constexpr (MODE == SRLHIP_RNG_PHILOX) {
        r.p.k0 = rs.key[e]; r.p.k1 = rs.key[n + e]; r.p.ctr = rs.ctr[e]; r.p.stream = 0;
    } else {
        r.m.load(rs.mt, e);
    }
}
template <int MODE>
__device__ __forceinline__ void rng_store(const typename RngSel<MODE>::type &r, const RngState &rs, int e) {
    if constexpr (MODE == SRLHIP_RNG_PHILOX) rs.ctr[e] = r.p.ctr;
    else if constexpr (MODE == SRLHIP_RNG_MT19937) r.m.store(rs.mt, e);
}

// reset: mobile_robot_env.py:166-181 and the variant overrides -------------------
template <class R>
__device__ __forceinline__ void reset_env(const MobileParams &p, R &rng, MobileEnv &m) {
    const double max_x = 4.0, max_y = 4.0;
    m.cur = 0;
    m.x = max_x / 2 + rng.uniform(-max_x / 3, max_x / 3);
    m.y = 0.0;
    if (p.kind != SRLHIP_ENV_MOBILE_1D) m.y = max_y / 2 + rng.uniform(-max_y / 3, max_y / 3);
    const double margin = 0.1 * max_x;
    m.tx = 0.9 * max_x; m.ty = 0.0; m.t2x = 0.0; m.t2y = 0.0;
    if (p.kind == SRLHIP_ENV_MOBILE_1D) {
        if (p.random_target) m.tx = rng.uniform(0 + margin, max_x - margin);
    } else if (p.kind == SRLHIP_ENV_MOBILE_LINE) {
        if (p.random_target) m.tx = rng.uniform(0 + margin, max_x - margin);
        m.ty = max_x;
    } else {
        m.ty = max_y * 3 / 4;
        if (p.random_target) {
            m.tx = rng.uniform(0 + margin, max_x - margin);
            m.ty = rng.uniform(0 + margin, max_y - margin);
        }
        if (p.kind == SRLHIP_ENV_MOBILE_2TARGET) {
            m.t2x = 0.1 * max_x; m.t2y = max_y * 3 / 4;
            if (p.random_target) {
                m.t2x = rng.uniform(0 + margin, max_x - margin);
                m.t2y = rng.uniform(0 + margin, max_y - margin);
            }
        }
    }
    m.counter = 0;
}

// getSRLState for ground_truth: getGroundTruth() - getTargetPos() (srl_env.py:39-42)
__device__ __forceinline__ void observe(const MobileParams &p, const MobileEnv &m, float &o0, float &o1) {
    double tx = m.cur ? m.t2x : m.tx, ty = m.cur ? m.t2y : m.ty;
    if (p.kind == SRLHIP_ENV_MOBILE_LINE) {          // 1-vector target broadcast over (x, y)
        double t = tx - 0.2;
        o0 = (float)(m.x - t); o1 = (float)(m.y - t);
    } else {
        o0 = (float)(m.x - tx); o1 = (float)(m.y - ty);
    }
}

// step: mobile_robot_env.py:235-280 --------------------------------------------
// KIND / DISC / SHAPE >= 0: env kind / action type / shaped reward known at compile time (rollout kernels); -1: read from the params.
// (SHAPE matters: with a run-time flag the compiler evaluates the shaped reward's float64 square root — a reciprocal-square-root
//  estimate and ten dependent FMAs — in every step and selects afterwards.)
template <int KIND = -1, int DISC = -1, int SHAPE = -1>
__device__ __forceinline__ void step_env(const MobileParams &pp, MobileEnv &m, int a, float a0, float a1, double dv,
                                         double &reward, bool &done) {
    struct { int32_t kind, is_discrete, shape_reward; } p = {KIND >= 0 ? KIND : pp.kind, DISC >= 0 ? DISC : pp.is_discrete,
                                                             SHAPE >= 0 ? SHAPE : pp.shape_reward};
    double dx = 0.0, dy = 0.0;
    if (p.is_discrete) {
        if (p.kind == SRLHIP_ENV_MOBILE_1D) {
            dx = a == 0 ? -dv : a == 1 ? dv : 0.0;
        } else {
            dx = a == 0 ? -dv : a == 1 ? dv : 0.0;
            dy = a == 2 ? -dv : a == 3 ? dv : 0.0;
        }
    } else {
        // float32 Box action * python-float dv: numpy keeps float32 (value-based casting)
        float fdv = (float)dv;
        dx = (double)(fmaxf(fminf(a0, 1.0f), -1.0f) * fdv);
        dy = (double)(fmaxf(fminf(a1, 1.0f), -1.0f) * fdv);
    }
    const double px = m.x, py = m.y;
    m.x = m.x + dx;
    if (p.kind != SRLHIP_ENV_MOBILE_1D) m.y = m.y + dy;
    // collision with the arena walls: x first, `break` on the first violation
    const double margin_x = 0.1 + (0.325 * 2) / 2, margin_y = 0.1 + 0.2 / 2;
    bool bumped = false;
    if (m.x < margin_x || m.x > 4 - margin_x) {
        bumped = true;
    } else if (p.kind != SRLHIP_ENV_MOBILE_1D && (m.y < margin_y || m.y > 4 - margin_y)) {
        bumped = true;
    }
    if (bumped) { m.x = px; m.y = py; }
    m.counter += 1;
    // _reward
    double tx = m.cur ? m.t2x : m.tx, ty = m.cur ? m.t2y : m.ty;
    double distance = 0.0;
    bool within;
    if (p.kind == SRLHIP_ENV_MOBILE_LINE) {
        distance = fabs((tx - 0.2) - m.x);
        within = distance <= 0.1;
    } else {
        // np.linalg.norm = sqrt(ddot): sqrt is correctly rounded and monotonic, so `sqrt(s2) <= 0.4` holds exactly when
        // s2 <= the largest double whose rounded square root is <= 0.4.  The sparse reward only needs that predicate;
        // the square root itself (a ~25-instruction dependent chain in f64) is taken only for the shaped reward.
        constexpr double kSqMax04 = 0x1.47ae147ae147cp-3;
        const double ex = tx - m.x, ey = ty - m.y;
        const double s2 = p.kind == SRLHIP_ENV_MOBILE_1D ? fma(ex, ex, 0.0) : fma(ey, ey, fma(ex, ex, 0.0));
        if (p.shape_reward) { distance = sqrt(s2); within = distance <= 0.4; }
        else within = s2 <= kSqMax04;
    }
    reward = 0.0;
    if (within) {
        reward = 1.0;
        if (p.kind == SRLHIP_ENV_MOBILE_2TARGET && m.cur < 1) m.cur += 1;
    }
    if (bumped) reward = -1.0;
    if (p.shape_reward) reward = -distance;
    done = m.counter > 250;                           // terminated is never set (:336-343)
}

__device__ __forceinline__ void account(const EpisodeStats &st, int e, double reward, bool done) {
    double r = st.ep_return[e] + reward;
    int32_t l = st.ep_length[e] + 1;
    st.last_reward[e] = reward;
    if (done) {
        st.last_return[e] = r; st.last_length[e] = l; st.n_finished[e] += 1;
        r = 0.0; l = 0;
    }
    st.ep_return[e] = r; st.ep_length[e] = l;
}

template <int MODE>
__global__ void __launch_bounds__(kBlock)
mobile_reset_k(MobileParams p, MobileState s, RngState rs, EpisodeStats st, const uint8_t *mask,
               const double *host_rand, int rand_stride, float *obs) {
    int e = blockIdx.x * kBlock + threadIdx.x;
    if (e >= p.n) return;
    if (mask && !mask[e]) return;
    typename RngSel<MODE>::type rng;
    rng_load<MODE>(rng, rs, e, p.n, host_rand, rand_stride, nullptr);
    MobileEnv m;
    load_env(s, e, m);
    reset_env(p, rng, m);
    store_env(s, e, m);
    rng_store<MODE>(rng, rs, e);
    st.ep_return[e] = 0.0; st.ep_length[e] = 0;
    if (obs) {
        float o0, o1;
        observe(p, m, o0, o1);
        if (p.kind == SRLHIP_ENV_MOBILE_1D) obs[e] = o0;
        else reinterpret_cast<float2 *>(obs)[e] = make_float2(o0, o1);
    }
}

// Synthetic random-agent action (rl_baselines/random_agent.py:36) from Philox stream 1 of the env.  Discrete: action i is word i % 4 of
// block i / 4 (multiply-shift into [0, m]) — FOUR actions per Philox block (round 6: the sampler's 131 072 blocks per 4096 x 2048
// rollout were 27 us of the whole chip beside a 25 us recurrence; oracle/mobile_oracle.c draws the same way); continuous: one block per
// action (two float64 uniforms).  `actr` counts ACTIONS.
__device__ __forceinline__ int discrete_from_word(const MobileParams &p, uint32_t w) {
    const uint32_t m = p.kind == SRLHIP_ENV_MOBILE_1D ? 1u : 3u;
    return (int)(uint32_t)(((uint64_t)w * ((uint64_t)m + 1)) >> 32);
}
__device__ __forceinline__ void sample_action(const MobileParams &p, uint32_t k0, uint32_t k1, uint64_t actr,
                                              int &a, float &a0, float &a1) {
    Philox ph; ph.k0 = k0; ph.k1 = k1; ph.stream = 1;
    uint32_t o[4];
    if (p.is_discrete) {
        ph.ctr = actr >> 2; ph.block(o);
        a = discrete_from_word(p, o[actr & 3]);
    } else {
        ph.ctr = actr; ph.block(o);
        a0 = (float)(-1.0 + 2.0 * Philox::to_double(o[0], o[1]));
        a1 = (float)(-1.0 + 2.0 * Philox::to_double(o[2], o[3]));
    }
}

// All T x N synthetic actions of a rollout at once, into a [T][N] action plane whose row 0 is action base[e] + offset of env e's stream.
// The plane is covered by `plane_rows` rows of `bpr` workgroups — continuous actions: row t, one block per entry; discrete actions: row q
// is the q-th Philox BLOCK the env's T actions touch (they start anywhere inside their first block: T / 4 + 1 blocks at most), the thread
// writes the up to four entries it yields.  Workgroup `sb` finds its row without an integer division (a 64-bit i / n and i % n per entry
// cost three times the Philox block they index): row = trunc(sb * (1 / bpr)) in float64, off by at most one, corrected.
struct PlaneMap { uint32_t bpr; double inv_bpr; };
inline PlaneMap plane_map(int64_t n) { PlaneMap m; m.bpr = (uint32_t)((n + kBlock - 1) / kBlock); m.inv_bpr = 1.0 / (double)m.bpr; return m; }
inline int64_t plane_rows(const MobileParams &p, int T) { return p.is_discrete ? (int64_t)(T + 3) / 4 + 1 : (int64_t)T; }
__device__ __forceinline__ void sample_plane_entry(const MobileParams &p, const uint32_t *key, const uint64_t *base, uint64_t offset, uint32_t sb,
                                                   const PlaneMap &pm, int T, void *__restrict__ act) {
    int32_t t = (int32_t)((double)sb * pm.inv_bpr);
    int32_t r = (int32_t)sb - t * (int32_t)pm.bpr;
    if (r < 0) { t -= 1; r += (int32_t)pm.bpr; } else if (r >= (int32_t)pm.bpr) { t += 1; r -= (int32_t)pm.bpr; }
    const int64_t e64 = (int64_t)r * kBlock + threadIdx.x;
    if (e64 >= p.n) return;
    const int e = (int)e64;
    if (p.is_discrete) {
        if (t > (T + 3) / 4) return;
        const uint64_t first = base[e] + offset;                    // the env's action index of plane row 0
        Philox ph; ph.k0 = key[e]; ph.k1 = key[p.n + e]; ph.stream = 1; ph.ctr = (first >> 2) + (uint64_t)t;
        const int64_t row0 = (int64_t)(ph.ctr * 4 - first);         // plane row of the block's word 0 (-3 .. T + 3)
        if (row0 >= T) return;
        uint32_t o[4]; ph.block(o);
#pragma unroll
        for (int w = 0; w < 4; w++) {
            const int64_t row = row0 + w;
            if (row >= 0 && row < T) static_cast<int32_t *>(act)[row * p.n + e] = discrete_from_word(p, o[w]);
        }
        return;
    }
    if (t >= T) return;
    const int64_t i = (int64_t)t * p.n + e;
    int a = 0; float a0 = 0.f, a1 = 0.f;
    sample_action(p, key[e], key[p.n + e], base[e] + offset + (uint64_t)t, a, a0, a1);
    static_cast<float2 *>(act)[i] = make_float2(a0, a1);
}
__global__ void __launch_bounds__(kBlock)
mobile_sample_actions_k(MobileParams p, RngState rs, int T, PlaneMap pm, void *__restrict__ act) {
    sample_plane_entry(p, rs.key, rs.act_ctr, 0, blockIdx.x, pm, T, act);
}

// One launch == T consecutive VecEnv steps; T == 1 with plain stores is the per-step entry point, T > 1 the fused
// rollout.  Actions always come from the [T][N] plane (the caller's, or the one mobile_sample_actions_k just filled:
// `advance_actr` then moves the env's action-stream counter past the T consumed blocks).
template <int MODE, int KIND, int DISC>
__global__ void __launch_bounds__(kBlock)
mobile_rollout_k(MobileParams p, MobileState s, RngState rs, EpisodeStats st, int T, const void *__restrict__ actions,
                 const double *__restrict__ noise, float *__restrict__ obs, float *__restrict__ rew,
                 uint8_t *__restrict__ done_out, int advance_actr, PersistArgs sig) {
    int e = blockIdx.x * kBlock + threadIdx.x;
    if (e >= p.n) return;
    if (KIND >= 0) p.kind = KIND;                 // compile-time constants from here on
    if (DISC >= 0) p.is_discrete = DISC;
    typename RngSel<MODE>::type rng;
    rng_load<MODE>(rng, rs, e, p.n, nullptr, 0, noise);
    MobileEnv m;
    load_env(s, e, m);
    // (Monitor's record of the last finished episode is written only when an episode finishes in this launch and is never read: on
    //  host-pointer handles those two planes are mapped host memory — srlhip_episode_records — and a read / an unconditional write-back
    //  would cross PCIe in every launch)
    double ep_ret = st.ep_return[e], last_ret = 0.0, last_reward = 0.0;
    int32_t ep_len = st.ep_length[e], last_len = 0, n_fin = st.n_finished[e];
    const int32_t n_fin0 = n_fin;
    const int32_t *act_i = static_cast<const int32_t *>(actions);
    const float2 *act_f = static_cast<const float2 *>(actions);
    // Actions are fetched kChunk steps at a time: on gfx9 loads and stores share one in-order counter (vmcnt), so waiting
    // for a load also drains every output store issued before it.  One wait per kChunk steps instead of one per step
    // keeps the streamed stores in flight.
    constexpr int kChunk = 16;
    for (int t0 = 0; t0 < T; t0 += kChunk) {
        int ai[kChunk]; float2 af[kChunk];
#pragma unroll
        for (int k = 0; k < kChunk; k++) {
            const int64_t r = (int64_t)min(t0 + k, T - 1) * p.n + e;
            if (p.is_discrete) ai[k] = act_i[r]; else af[k] = act_f[r];
        }
        // one full drain per chunk (vmcnt(0), expcnt / lgkmcnt untouched): afterwards no step of the chunk waits on memory
        __builtin_amdgcn_s_waitcnt(0x0F70);
#pragma unroll
        for (int k = 0; k < kChunk; k++) {
            const int t = t0 + k;
            if (t >= T) continue;                     // (uniform) tail of the last chunk
            const int64_t row = (int64_t)t * p.n + e;
            const int a = p.is_discrete ? ai[k] : 0;
            const float a0 = p.is_discrete ? 0.f : af[k].x, a1 = p.is_discrete ? 0.f : af[k].y;
            double dv = 0.1 + rng.normal(0.0, 0.0);       // DELTA_POS + N(0, NOISE_STD = 0): drawn, value 0
            double reward; bool done;
            step_env<KIND, DISC>(p, m, a, a0, a1, dv, reward, done);
            ep_ret += reward; ep_len += 1; last_reward = reward;
            if (done) {
                last_ret = ep_ret; last_len = ep_len; n_fin += 1; ep_ret = 0.0; ep_len = 0;
                if (p.auto_reset) reset_env(p, rng, m);
            }
            float o0, o1;
            observe(p, m, o0, o1);
            if (obs) {
                if (p.kind == SRLHIP_ENV_MOBILE_1D) __builtin_nontemporal_store(o0, obs + row);
                else {
                    __builtin_nontemporal_store(f32x2{o0, o1}, reinterpret_cast<f32x2 *>(obs + 2 * row));       // one 8-byte store per lane: a wavefront row is 512 contiguous bytes
                }
            }
            if (rew) __builtin_nontemporal_store((float)reward, rew + row);
            if (done_out) __builtin_nontemporal_store((uint8_t)done, done_out + row);
        }
    }
    if (sig.done) {
        // EARLY COMPLETION SIGNAL of a single-step launch on a host-pointer handle (api.hip host_step_begin; the Kuka kernels'
        // kuka_tree_kernels.hpp has the long version): the step's outputs are in this XCD's L2; per eighth of the grid (workgroups
        // b = g mod 8: one XCD, checked through the eighth's XCD tag) the last WAVEFRONT to arrive writes that L2 back and posts the
        // step's sequence number — the host does not wait for the exit stores below, the kernel's end and the stream synchronisation.
        if (n_fin != n_fin0) {           // Monitor's record: the host reads it right after the step
            __hip_atomic_store(st.last_return + e, last_ret, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(st.last_length + e, last_len, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        const int g8 = (int)blockIdx.x & 7, waves = (p.n + 63) / 64;
        int real = 0;                    // wavefronts with a live lane in the workgroups of this eighth
        for (int b = g8; b * (kBlock / 64) < waves; b += 8) real += min(kBlock / 64, waves - b * (kBlock / 64));
        uint32_t xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        uint32_t *cnt = sig.count + g8 * kPersistWordStride, *tag = cnt + 1;
        const bool first = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0;       // the wavefront's first ACTIVE lane
        if (first) __hip_atomic_fetch_or(tag, 1u << (xcc & 15u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_s_waitcnt(0x0F70);
        asm volatile("" ::: "memory");
        uint32_t last = 0;
        if (first) last = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == (uint32_t)real * sig.start_seq;
        if (__builtin_amdgcn_readfirstlane(last)) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
            if (first) {
                const uint32_t seen = __hip_atomic_exchange(tag, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(sig.done + g8, (seen & (seen - 1u)) ? ~sig.start_seq : sig.start_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
    store_env(s, e, m);
    rng_store<MODE>(rng, rs, e);
    if (advance_actr) rs.act_ctr[e] += (uint64_t)T;
    st.ep_return[e] = ep_ret; st.ep_length[e] = ep_len;
    if (n_fin != n_fin0) { st.last_return[e] = last_ret; st.last_length[e] = last_len; }
    st.n_finished[e] = n_fin; st.last_reward[e] = last_reward;
}

// Persistent stepping (srlhip_set_persistent; the protocol of the Kuka kernels, kuka_tree_kernels.hpp): ONE launch stays resident with every
// env's state in registers and takes its steps from the host through mapped memory — workgroup 0 polls the host's sequence number and
// relays it, the others poll the relay; every lane reads its action from the mapped plane, steps, stores its outputs straight to the
// host's mapped planes (plain stores: they stay in the XCD's L2), and the last wavefront of each eighth of the grid (workgroups b = g mod
// 8: one XCD, verified behind a start barrier) writes that L2 back and posts the eighth's `done` word.  Where an eighth does not sit on
// one XCD the outputs are written through instead (system-scope stores: slow, correct).  Same step code as mobile_rollout_k: bit-identical.
template <int MODE, int KIND, int DISC>
__global__ void __launch_bounds__(kBlock)
mobile_persist_k(MobileParams p, MobileState s, RngState rs, EpisodeStats st, const void *actions, float *obs, float *rew, uint8_t *done_out, PersistArgs pa) {
    __shared__ uint32_t tok_s, verdict_s;
    const int e = blockIdx.x * kBlock + threadIdx.x;
    const bool valid = e < p.n;
    const int ee = valid ? e : p.n - 1;
    p.kind = KIND; p.is_discrete = DISC;
    uint32_t xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) __hip_atomic_fetch_add(pa.ctrl, 1u + (((xcc & 15u) != (blockIdx.x & 7u)) ? 0x10000u : 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    typename RngSel<MODE>::type rng;
    rng_load<MODE>(rng, rs, ee, p.n, nullptr, 0, nullptr);
    MobileEnv m;
    load_env(s, ee, m);
    double ep_ret = st.ep_return[ee], last_ret = 0.0, last_reward = st.last_reward[ee];
    int32_t ep_len = st.ep_length[ee], last_len = 0, n_fin = st.n_finished[ee];
    // start barrier: every workgroup has registered -> 1: direct outputs, 2: written-through outputs, 3: told to stop while waiting
    if (threadIdx.x == 0) {
        uint32_t verdict = 0;
        uint32_t *vw = pa.ctrl + kPersistWordStride;
        if (blockIdx.x == 0) {
            for (;;) {
                const uint32_t reg = __hip_atomic_load(pa.ctrl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((reg & 0xffffu) == gridDim.x) { verdict = ((reg >> 16) || pa.force_staged) ? 2u : 1u; break; }
                if (__hip_atomic_load(pa.stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) { verdict = 3u; break; }
                __builtin_amdgcn_s_sleep(8);
            }
            __hip_atomic_store(vw, verdict, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            while (!(verdict = __hip_atomic_load(vw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) __builtin_amdgcn_s_sleep(8);
        }
        verdict_s = verdict;
    }
    __syncthreads();
    const uint32_t verdict = verdict_s;
    const bool direct = verdict == 1u;
    const int g8 = (int)blockIdx.x & 7, waves = (p.n + 63) / 64;
    int real = 0;                    // wavefronts with a live lane in the workgroups of this eighth
    for (int b = g8; b * (kBlock / 64) < waves; b += 8) real += min(kBlock / 64, waves - b * (kBlock / 64));
    const bool first = valid && __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0;     // (lanes of a wavefront are valid from lane 0 up)
    uint32_t my_seq = pa.start_seq, k = 0;
    const int od = p.kind == SRLHIP_ENV_MOBILE_1D ? 1 : 2;
    while (verdict != 3u) {
        if (threadIdx.x == 0) {
            uint32_t token;
            if (blockIdx.x == 0) {
                uint32_t sq = my_seq, stop = 0, spins = 0;
                for (;;) {
                    const uint64_t w = __hip_atomic_load(reinterpret_cast<const uint64_t *>(pa.seq), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    sq = (uint32_t)w; stop = (uint32_t)(w >> 32);
                    if (sq != my_seq || stop || ++spins >= pa.spin_limit) break;
                    __builtin_amdgcn_s_sleep(8);
                }
                token = sq != my_seq ? sq : kPersistPark;
                if (token == kPersistPark) __hip_atomic_store(pa.parked, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                for (int x = 0; x < 8; x++) __hip_atomic_store(pa.relay + x * kPersistWordStride, token, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                const uint32_t *rw = pa.relay + (blockIdx.x & 7) * kPersistWordStride;
                while ((token = __hip_atomic_load(rw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == my_seq) __builtin_amdgcn_s_sleep(8);
            }
            tok_s = token;
        }
        __syncthreads();
        const uint32_t token = tok_s;
        __syncthreads();                                   // everybody has read the token before thread 0 writes the next one
        if (token == kPersistPark) break;
        my_seq = token; k += 1;
        if (valid) {
            int a = 0; float a0 = 0.f, a1 = 0.f;
            if (p.is_discrete) a = __hip_atomic_load(static_cast<const int32_t *>(actions) + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            else {
                a0 = __hip_atomic_load(static_cast<const float *>(actions) + 2 * e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                a1 = __hip_atomic_load(static_cast<const float *>(actions) + 2 * e + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            double dv = 0.1 + rng.normal(0.0, 0.0);       // DELTA_POS + N(0, NOISE_STD = 0): drawn, value 0
            double reward; bool done;
            step_env<KIND, DISC>(p, m, a, a0, a1, dv, reward, done);
            ep_ret += reward; ep_len += 1; last_reward = reward;
            if (done) {
                last_ret = ep_ret; last_len = ep_len; n_fin += 1; ep_ret = 0.0; ep_len = 0;
                if (p.auto_reset) reset_env(p, rng, m);
                __hip_atomic_store(st.last_return + e, last_ret, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);        // Monitor's record, with the step
                __hip_atomic_store(st.last_length + e, last_len, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            float o0, o1;
            observe(p, m, o0, o1);
            if (direct) {
                if (od == 1) obs[e] = o0; else { obs[2 * e] = o0; obs[2 * e + 1] = o1; }
                rew[e] = (float)reward; done_out[e] = (uint8_t)done;
            } else {
                __hip_atomic_store(obs + od * e, o0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if (od == 2) __hip_atomic_store(obs + 2 * e + 1, o1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(rew + e, (float)reward, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(done_out + e, (uint8_t)done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);
        asm volatile("" ::: "memory");
        uint32_t last = 0;
        if (first) last = __hip_atomic_fetch_add(pa.count + g8 * kPersistWordStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == (uint32_t)real * k;
        if (__builtin_amdgcn_readfirstlane(last)) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
            if (first) __hip_atomic_store(pa.done + g8, my_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    if (!valid) return;
    store_env(s, e, m);
    rng_store<MODE>(rng, rs, e);
    st.ep_return[e] = ep_ret; st.ep_length[e] = ep_len;
    st.n_finished[e] = n_fin; st.last_reward[e] = last_reward;
}

// Episode-parallel rollout (Philox mode with auto-reset).  In the MobileRobot family an episode always lasts exactly
// 251 steps (`done = counter > 250`, `terminated` is never set: mobile_robot_env.py:336-343) and a reset consumes a
// fixed number of counter-based draws, so the state an env has right after its k-th reset inside a rollout is a pure
// function of (key, counter + (k-1) * draws): the T-step rollout of one env splits into independent segments — the
// rest of its running episode, then whole episodes — which run on separate lanes.  At the 4096-env BASELINE size a
// 2048-step rollout becomes 4096 x 10 lanes of <= 251 steps instead of 4096 lanes of 2048 steps: the recurrence is the
// only sequential thing on this path, and this cuts it ~8x.  Outputs, final state, counters and episode statistics
// are bit-identical to mobile_rollout_k (tests/test_gpu_mobile.py).  Lane (j, e) = segment j of env e; the waves of a
// segment slot store whole [t][e..e+63] rows when their envs' episode clocks agree (they do after a common reset).
constexpr int kEpisodeSteps = 251;

// Everything a segment lane reads from the per-env state.  The lanes of one env are NOT guaranteed to be co-resident
// (beyond ~0.5 M lanes later blocks are dispatched after earlier ones retire), and the lane that finishes an env's rollout
// overwrites its state: the segments therefore read this snapshot, taken by mobile_snapshot_k in stream order right
// before the launch, and only write the live state.
struct MobileSnap { MobileState s; const uint64_t *ctr; const double *ep_return; const int32_t *ep_length; };
// where the lane that leaves an env's final state ALSO writes it: the snapshot set the next launch will read (null planes: nowhere)
struct MobileSnapOut { MobileState s; uint64_t *ctr, *actr; double *ep_return; int32_t *ep_length; };
// The synthetic agent's NEXT action plane, drawn by the workgroups of a rollout launch beyond its segment lanes (the rollout itself
// is a latency-bound recurrence on 10 x N lanes: the chip has room) so that the following rollout starts without a sampler launch:
// block base[e] + T + t of env e's action stream (base = the counters in the snapshot: the live ones move when the rollout ends).
struct NextPlane { void *act; const uint64_t *base; int ep_blocks; PlaneMap pm; };

__global__ void __launch_bounds__(kBlock)
mobile_snapshot_k(int n, MobileState s, RngState rs, EpisodeStats st, MobileState d, uint64_t *ctr, double *ep_return, int32_t *ep_length, uint64_t *actr) {
    const int e = blockIdx.x * kBlock + threadIdx.x;
    if (e >= n) return;
    actr[e] = rs.act_ctr[e];
    d.pos_x[e] = s.pos_x[e]; d.pos_y[e] = s.pos_y[e]; d.tgt_x[e] = s.tgt_x[e]; d.tgt_y[e] = s.tgt_y[e];
    d.tgt2_x[e] = s.tgt2_x[e]; d.tgt2_y[e] = s.tgt2_y[e]; d.counter[e] = s.counter[e]; d.cur_target[e] = s.cur_target[e];
    ctr[e] = rs.ctr[e]; ep_return[e] = st.ep_return[e]; ep_length[e] = st.ep_length[e];
}

template <int KIND, int DISC, int SHAPE, int PLANES>      // PLANES: 0 check every output plane, 1 obs / reward / done present, 2 those and act_out
__global__ void __launch_bounds__(kBlock)
mobile_rollout_ep_k(MobileParams p, MobileState s, MobileSnap snap, RngState rs, EpisodeStats st, int T, int draws_per_reset, int smax,
                    const void *__restrict__ actions, float *__restrict__ obs, float *__restrict__ rew,
                    uint8_t *__restrict__ done_out, int advance_actr, NextPlane next, void *__restrict__ act_out, MobileSnapOut snap_out) {
    p.kind = KIND; p.is_discrete = DISC; p.shape_reward = SHAPE;            // compile-time constants from here on
    if ((int)blockIdx.x >= next.ep_blocks) {                               // spare workgroups: the next rollout's action plane
        sample_plane_entry(p, rs.key, next.base, (uint64_t)T, blockIdx.x - (uint32_t)next.ep_blocks, next.pm, T, next.act);
        return;
    }
    const int64_t gid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int e = (int)(gid % p.n), j = (int)(gid / p.n);
    if (j >= smax) return;
    const int c0 = snap.s.counter[e];
    const int L0 = c0 <= kEpisodeSteps - 1 ? kEpisodeSteps - c0 : 1;       // steps until the running episode ends
    const int t_lo = j == 0 ? 0 : L0 + kEpisodeSteps * (j - 1);
    if (t_lo >= T) return;
    const int t_end = L0 + kEpisodeSteps * j;                             // the step after this segment's episode ends
    const int t_hi = t_end < T ? t_end : T;
    const bool last = t_hi == T;                                          // this lane leaves the env's final state
    const int n_completed = T >= L0 ? 1 + (T - L0) / kEpisodeSteps : 0;   // episodes of env e that finish inside the rollout
    PhiloxRng rng;
    rng.p.k0 = rs.key[e]; rng.p.k1 = rs.key[p.n + e]; rng.p.stream = 0;
    const uint64_t ctr0 = snap.ctr[e];
    MobileEnv m;
    double ep_ret = 0.0, seg_ret = 0.0, last_reward = 0.0;
    int32_t ep_len = 0, seg_len = 0;
    if (j == 0) {
        load_env(snap.s, e, m);
        rng.p.ctr = ctr0;
        ep_ret = snap.ep_return[e]; ep_len = snap.ep_length[e];
    } else {
        rng.p.ctr = ctr0 + (uint64_t)(j - 1) * (uint64_t)draws_per_reset;
        reset_env(p, rng, m);                                              // the reset the previous segment ends with
    }
    const int32_t *act_i = static_cast<const int32_t *>(actions);
    const float2 *act_f = static_cast<const float2 *>(actions);
    constexpr int kChunk = 16;                                             // see mobile_rollout_k: one vmcnt drain per chunk
    int t_start = t_lo;
    // Fast loop: whole chunks of INTERIOR steps.  Inside a segment `done` can only fire at the step that ends its episode
    // (t_end - 1), so every earlier step needs no reset; with all four output planes present (ALL) and every lane of the wavefront
    // holding a full chunk, a chunk is one straight-line block: no per-step predicate, no reset body to branch around, no plane
    // checks — the scheduler overlaps one step's conversions and stores with the next step's dependent chain.  Ragged clocks, the
    // last partial chunk and the episode-ending step go through the general loop below.
    if constexpr (PLANES > 0) {
        // The actions of the NEXT chunk are requested before this chunk's steps: loads and stores retire in order on one counter
        // (vmcnt), so a load issued ahead of the chunk's 3-4 x kFast stores is awaited with those stores still in flight (the compiler
        // counts them: the block is straight-line) instead of exposing a full memory round trip per chunk.  8 + 32 operations stay
        // below the counter's 63.
        constexpr int kFast = 8;
        const int t_int_end = t_end - 1 < t_hi ? t_end - 1 : t_hi;       // interior steps: [t_lo, t_int_end)
        int ai[kFast]; float2 af[kFast];
        auto fetch = [&](int t0, int *di, float2 *df) {
#pragma unroll
            for (int k = 0; k < kFast; k++) {
                const int64_t r = (int64_t)min(t0 + k, t_hi - 1) * p.n + e;   // clamped: a prefetch past the segment is harmless
                if (p.is_discrete) di[k] = act_i[r]; else df[k] = act_f[r];
            }
        };
        auto chunk = [&](const int *ci, const float2 *cf) {
#pragma unroll
            for (int k = 0; k < kFast; k++) {
                const int64_t row = (int64_t)(t_start + k) * p.n + e;
                const int a = p.is_discrete ? ci[k] : 0;
                const float a0 = p.is_discrete ? 0.f : cf[k].x, a1 = p.is_discrete ? 0.f : cf[k].y;
                if constexpr (PLANES == 2) {
                    if (p.is_discrete) __builtin_nontemporal_store(a, static_cast<int32_t *>(act_out) + row);
                    else static_cast<float2 *>(act_out)[row] = make_float2(a0, a1);
                }
                const double dv = 0.1 + rng.normal(0.0, 0.0);
                double reward; bool done;
                step_env<KIND, DISC, SHAPE>(p, m, a, a0, a1, dv, reward, done);     // done is false here (interior step)
                ep_ret += reward; ep_len += 1; last_reward = reward;
                float o0, o1;
                observe(p, m, o0, o1);
                if (p.kind == SRLHIP_ENV_MOBILE_1D) __builtin_nontemporal_store(o0, obs + row);
                else __builtin_nontemporal_store(f32x2{o0, o1}, reinterpret_cast<f32x2 *>(obs + 2 * row));
                __builtin_nontemporal_store((float)reward, rew + row);
                __builtin_nontemporal_store((uint8_t)0, done_out + row);
            }
            t_start += kFast;
        };
        if (__all(t_start + kFast <= t_int_end)) {
            fetch(t_start, ai, af);
            __builtin_amdgcn_s_waitcnt(0x0F70);                            // vmcnt(0) once, so that the loop header has nothing pending on either path
            do {
                int ni[kFast]; float2 nf[kFast];
                fetch(t_start + kFast, ni, nf);
                __builtin_amdgcn_sched_barrier(0);                         // keep the requests AHEAD of the chunk's stores
                chunk(ai, af);
                __builtin_amdgcn_sched_barrier(0);
                // the copies are the first use of the prefetched registers: the wait lands HERE, where what is pending is known
                // exactly (8 loads, then 32 stores) — at the loop header it would be merged with the entry path and drain everything
#pragma unroll
                for (int k = 0; k < kFast; k++) { ai[k] = ni[k]; af[k] = nf[k]; }
            } while (__all(t_start + kFast <= t_int_end));
        }
    }
    for (int t0 = t_start; t0 < t_hi; t0 += kChunk) {
        int ai[kChunk]; float2 af[kChunk];
#pragma unroll
        for (int k = 0; k < kChunk; k++) {
            const int64_t r = (int64_t)min(t0 + k, t_hi - 1) * p.n + e;
            if (p.is_discrete) ai[k] = act_i[r]; else af[k] = act_f[r];
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);
#pragma unroll
        for (int k = 0; k < kChunk; k++) {
            const int t = t0 + k;
            if (t < t_hi) {
                const int64_t row = (int64_t)t * p.n + e;
                const int a = p.is_discrete ? ai[k] : 0;
                const float a0 = p.is_discrete ? 0.f : af[k].x, a1 = p.is_discrete ? 0.f : af[k].y;
                if (act_out) {                                             // the plane is internal: hand the actions taken to the caller
                    if (p.is_discrete) __builtin_nontemporal_store(a, static_cast<int32_t *>(act_out) + row);
                    else static_cast<float2 *>(act_out)[row] = make_float2(a0, a1);
                }
                const double dv = 0.1 + rng.normal(0.0, 0.0);              // DELTA_POS + N(0, NOISE_STD = 0): draws nothing
                double reward; bool done;
                step_env<KIND, DISC, SHAPE>(p, m, a, a0, a1, dv, reward, done);
                ep_ret += reward; ep_len += 1; last_reward = reward;
                if (done) {
                    seg_ret = ep_ret; seg_len = ep_len; ep_ret = 0.0; ep_len = 0;
                    reset_env(p, rng, m);
                }
                float o0, o1;
                observe(p, m, o0, o1);
                if (obs) {
                    if (p.kind == SRLHIP_ENV_MOBILE_1D) __builtin_nontemporal_store(o0, obs + row);
                    else {
                        __builtin_nontemporal_store(f32x2{o0, o1}, reinterpret_cast<f32x2 *>(obs + 2 * row));       // one 8-byte store per lane: a wavefront row is 512 contiguous bytes
                    }
                }
                if (rew) __builtin_nontemporal_store((float)reward, rew + row);
                if (done_out) __builtin_nontemporal_store((uint8_t)done, done_out + row);
            }
        }
    }
    if (t_end <= T && j == n_completed - 1) {        // the last episode that finished inside the rollout
        st.last_return[e] = seg_ret; st.last_length[e] = seg_len;
    }
    if (last) {
        store_env(s, e, m);
        rs.ctr[e] = rng.p.ctr;
        const uint64_t actr = rs.act_ctr[e] + (advance_actr ? (uint64_t)T : 0);
        if (advance_actr) rs.act_ctr[e] = actr;
        st.ep_return[e] = ep_ret; st.ep_length[e] = ep_len; st.last_reward[e] = last_reward;
        st.n_finished[e] += n_completed;
        if (snap_out.ctr) {                  // the next launch's snapshot (this launch reads the other set)
            store_env(snap_out.s, e, m);
            snap_out.ctr[e] = rng.p.ctr; snap_out.actr[e] = actr; snap_out.ep_return[e] = ep_ret; snap_out.ep_length[e] = ep_len;
        }
    }
}

MobileParams params_of(const Handle *h) {
    MobileParams p;
    p.kind = h->cfg.env_kind; p.is_discrete = h->cfg.is_discrete; p.random_target = h->cfg.random_target;
    p.shape_reward = h->cfg.shape_reward; p.auto_reset = h->cfg.auto_reset; p.n = h->n;
    return p;
}

}  // namespace

int mobile_reset_rand_count(const srlhip_config &c) {
    int base = c.env_kind == SRLHIP_ENV_MOBILE_1D ? 1 : 2;
    if (!c.random_target) return base;
    switch (c.env_kind) {
        case SRLHIP_ENV_MOBILE_1D: case SRLHIP_ENV_MOBILE_LINE: return base + 1;
        case SRLHIP_ENV_MOBILE_2TARGET: return base + 4;
        default: return base + 2;
    }
}

int mobile_alloc(Handle *h) {
    MobileState &s = h->mobile;
    size_t n = (size_t)h->n;
    int rc = 0;
    if ((rc = h->dalloc(&s.pos_x, n)) || (rc = h->dalloc(&s.pos_y, n)) || (rc = h->dalloc(&s.tgt_x, n)) ||
        (rc = h->dalloc(&s.tgt_y, n)) || (rc = h->dalloc(&s.tgt2_x, n)) || (rc = h->dalloc(&s.tgt2_y, n)) ||
        (rc = h->dalloc(&s.counter, n)) || (rc = h->dalloc(&s.cur_target, n)))
        return rc;
    if (h->cfg.rng_mode == SRLHIP_RNG_PHILOX) {
        // snapshot planes of the episode-parallel rollout (launch_rollout_ep): allocated with the handle, never lazily — a
        // first rollout may sit inside srlhip_graph_begin / srlhip_graph_end, where hipMalloc is illegal
        MobileState &d = h->mobile_snap;
        if ((rc = h->dalloc(&d.pos_x, n)) || (rc = h->dalloc(&d.pos_y, n)) || (rc = h->dalloc(&d.tgt_x, n)) || (rc = h->dalloc(&d.tgt_y, n)) ||
            (rc = h->dalloc(&d.tgt2_x, n)) || (rc = h->dalloc(&d.tgt2_y, n)) || (rc = h->dalloc(&d.counter, n)) || (rc = h->dalloc(&d.cur_target, n)) ||
            (rc = h->dalloc(&h->snap_ep_return, n)) || (rc = h->dalloc(&h->snap_ep_length, n)) || (rc = h->dalloc(&h->snap_ctr, n)) ||
            (rc = h->dalloc(&h->snap_actr, n)))
            return rc;
        MobileState &d2 = h->mobile_snap2;
        if ((rc = h->dalloc(&d2.pos_x, n)) || (rc = h->dalloc(&d2.pos_y, n)) || (rc = h->dalloc(&d2.tgt_x, n)) || (rc = h->dalloc(&d2.tgt_y, n)) ||
            (rc = h->dalloc(&d2.tgt2_x, n)) || (rc = h->dalloc(&d2.tgt2_y, n)) || (rc = h->dalloc(&d2.counter, n)) || (rc = h->dalloc(&d2.cur_target, n)) ||
            (rc = h->dalloc(&h->snap2_ep_return, n)) || (rc = h->dalloc(&h->snap2_ep_length, n)) || (rc = h->dalloc(&h->snap2_ctr, n)) ||
            (rc = h->dalloc(&h->snap2_actr, n)))
            return rc;
    }
    return 0;
}

void mobile_free(Handle *) {}

int mobile_reset(Handle *h, const uint8_t *d_mask, const double *d_host_rand, float *d_obs) {
    h->snap_valid = false;
    MobileParams p = params_of(h);
    dim3 grid((h->n + kBlock - 1) / kBlock), block(kBlock);
    int stride = mobile_reset_rand_count(h->cfg);
    switch (h->cfg.rng_mode) {
        case SRLHIP_RNG_HOST:
            if (!d_host_rand) return h->fail(SRLHIP_EINVAL, "reset: RNG_HOST needs host_rand");
            hipLaunchKernelGGL(mobile_reset_k<SRLHIP_RNG_HOST>, grid, block, 0, h->stream, p, h->mobile, h->rng,
                               h->stats, d_mask, d_host_rand, stride, d_obs);
            break;
        case SRLHIP_RNG_PHILOX:
            hipLaunchKernelGGL(mobile_reset_k<SRLHIP_RNG_PHILOX>, grid, block, 0, h->stream, p, h->mobile, h->rng,
                               h->stats, d_mask, d_host_rand, stride, d_obs);
            break;
        default:
            hipLaunchKernelGGL(mobile_reset_k<SRLHIP_RNG_MT19937>, grid, block, 0, h->stream, p, h->mobile, h->rng,
                               h->stats, d_mask, d_host_rand, stride, d_obs);
    }
    SRL_HIP_CHECK(h, hipGetLastError());
    return 0;
}

namespace {

template <int MODE>
void launch_rollout(Handle *h, const MobileParams &p, int T, const void *d_actions, const double *d_noise, float *d_obs,
                    float *d_rew, uint8_t *d_done, int advance_actr) {
    dim3 grid((h->n + kBlock - 1) / kBlock), block(kBlock);
    PersistArgs sig{};                          // the early completion signal of a single-step launch with the caller's actions (api.hip)
    if (h->step_signal && T == 1 && d_actions && !advance_actr) {
        sig = *h->step_signal; h->step_signal_armed = true;
        h->signal_eighths = grid.x >= 8 ? 0xffu : (1u << grid.x) - 1u;      // eighth g = the workgroups b = g mod 8
    }
#define SRL_GO(KIND, DISC)                                                                                              \
    hipLaunchKernelGGL((mobile_rollout_k<MODE, KIND, DISC>), grid, block, 0, h->stream, p, h->mobile, h->rng, h->stats, T, \
                       d_actions, d_noise, d_obs, d_rew, d_done, advance_actr, sig)
#define SRL_KIND(KIND) { if (p.is_discrete) SRL_GO(KIND, 1); else SRL_GO(KIND, 0); }
    switch (p.kind) {
        case SRLHIP_ENV_MOBILE: SRL_KIND(SRLHIP_ENV_MOBILE) break;
        case SRLHIP_ENV_MOBILE_1D: SRL_KIND(SRLHIP_ENV_MOBILE_1D) break;
        case SRLHIP_ENV_MOBILE_2TARGET: SRL_KIND(SRLHIP_ENV_MOBILE_2TARGET) break;
        default: SRL_KIND(SRLHIP_ENV_MOBILE_LINE)
    }
#undef SRL_KIND
#undef SRL_GO
}

// next_plane: where the spare workgroups put the following rollout's actions (null: none); act_out: the caller's action plane
// when `d_actions` is an internal one
int launch_rollout_ep(Handle *h, const MobileParams &p, int T, const void *d_actions, float *d_obs, float *d_rew,
                      uint8_t *d_done, int advance_actr, void *next_plane, void *act_out) {
    // two snapshot sets in turn: this launch reads set `cur` — written by the previous rollout launch's final-state lanes, or by
    // mobile_snapshot_k now when anything else has touched the state since — and writes the env's final state into the other one.
    // Inside a stream capture the copy kernel always runs and nothing is written ahead: a replayed graph must not depend on which set
    // the previous replay left behind.
    hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(h->stream, &capture);
    const bool chained = capture == hipStreamCaptureStatusNone;
    const int cur = h->snap_cur;
    MobileState &sin = cur ? h->mobile_snap2 : h->mobile_snap, &sout = cur ? h->mobile_snap : h->mobile_snap2;
    uint64_t *in_ctr = cur ? h->snap2_ctr : h->snap_ctr, *in_actr = cur ? h->snap2_actr : h->snap_actr;
    double *in_ret = cur ? h->snap2_ep_return : h->snap_ep_return;
    int32_t *in_len = cur ? h->snap2_ep_length : h->snap_ep_length;
    if (!h->snap_valid || !chained)
        hipLaunchKernelGGL(mobile_snapshot_k, dim3((h->n + kBlock - 1) / kBlock), dim3(kBlock), 0, h->stream, h->n, h->mobile, h->rng, h->stats,
                           sin, in_ctr, in_ret, in_len, in_actr);
    const MobileSnap snap{sin, in_ctr, in_ret, in_len};
    MobileSnapOut snap_out = {};
    if (chained)
        snap_out = MobileSnapOut{sout, cur ? h->snap_ctr : h->snap2_ctr, cur ? h->snap_actr : h->snap2_actr,
                                 cur ? h->snap_ep_return : h->snap2_ep_return, cur ? h->snap_ep_length : h->snap2_ep_length};
    h->snap_cur = chained ? cur ^ 1 : cur;
    h->snap_valid = chained;
    const int smax = 1 + (T - 1 + kEpisodeSteps - 1) / kEpisodeSteps;     // first segment of one step + whole episodes
    const int64_t lanes = (int64_t)smax * h->n;
    const int ep_blocks = (int)((lanes + kBlock - 1) / kBlock);
    const PlaneMap pm = plane_map(h->n);
    const int64_t extra = next_plane ? plane_rows(p, T) * pm.bpr : 0;
    const NextPlane next{next_plane, in_actr, ep_blocks, pm};
    dim3 grid((unsigned)(ep_blocks + extra)), block(kBlock);
    const int draws = mobile_reset_rand_count(h->cfg);
    const int planes = d_obs && d_rew && d_done ? (act_out ? 2 : 1) : 0;   // the fast loop stores without checking
#define SRL_EP(KIND, DISC, SHAPE, PLANES)                                                                               \
    hipLaunchKernelGGL((mobile_rollout_ep_k<KIND, DISC, SHAPE, PLANES>), grid, block, 0, h->stream, p, h->mobile, snap, h->rng, h->stats, T, \
                       draws, smax, d_actions, d_obs, d_rew, d_done, advance_actr, next, act_out, snap_out)
#define SRL_PL(KIND, DISC, SHAPE) { if (planes == 2) SRL_EP(KIND, DISC, SHAPE, 2); else if (planes == 1) SRL_EP(KIND, DISC, SHAPE, 1); else SRL_EP(KIND, DISC, SHAPE, 0); }
#define SRL_GO(KIND, DISC) { if (p.shape_reward) SRL_PL(KIND, DISC, 1) else SRL_PL(KIND, DISC, 0) }
#define SRL_KIND(KIND) { if (p.is_discrete) SRL_GO(KIND, 1) else SRL_GO(KIND, 0) }
    switch (p.kind) {
        case SRLHIP_ENV_MOBILE: SRL_KIND(SRLHIP_ENV_MOBILE) break;
        case SRLHIP_ENV_MOBILE_1D: SRL_KIND(SRLHIP_ENV_MOBILE_1D) break;
        case SRLHIP_ENV_MOBILE_2TARGET: SRL_KIND(SRLHIP_ENV_MOBILE_2TARGET) break;
        default: SRL_KIND(SRLHIP_ENV_MOBILE_LINE)
    }
#undef SRL_KIND
#undef SRL_GO
#undef SRL_PL
#undef SRL_EP
    return 0;
}

}  // namespace

int mobile_rollout(Handle *h, int T, const void *d_actions, float *d_obs, float *d_rew, uint8_t *d_done,
                   void *d_act_out) {
    MobileParams p = params_of(h);
    if (h->cfg.rng_mode != SRLHIP_RNG_PHILOX && h->cfg.rng_mode != SRLHIP_RNG_MT19937)
        return h->fail(SRLHIP_EINVAL, "rollout: needs a device RNG mode (PHILOX or MT19937)");
    int advance = 0;
    const bool ep_path = h->cfg.rng_mode == SRLHIP_RNG_PHILOX && p.auto_reset && T >= 32;
    if (!d_actions && ep_path && !getenv("SRLHIP_NO_ACTION_PREFETCH")) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(h->stream, &cap);
        if (cap == hipStreamCaptureStatusNone) {
            // synthetic agent, episode-parallel rollout: two internal action planes in turn.  This rollout reads the plane the
            // previous launch's spare workgroups filled (or fills one now), its own spare workgroups fill the other one.
            const size_t bytes = (size_t)T * h->n * (p.is_discrete ? 4 : 8);
            for (int b = 0; b < 2; b++)
                if (h->act_plane_sz[b] < bytes) {
                    SRL_HIP_CHECK(h, hipStreamSynchronize(h->stream));
                    if (h->act_plane[b]) (void)hipFree(h->act_plane[b]);
                    h->act_plane[b] = nullptr; h->act_plane_sz[b] = 0; h->prefetch_valid = false;
                    SRL_HIP_CHECK(h, hipMalloc(&h->act_plane[b], bytes));
                    h->act_plane_sz[b] = bytes;
                }
            int cur = 0;
            if (h->prefetch_valid && h->prefetch_T == T) cur = h->prefetch_buf;
            else {
                const PlaneMap pm = plane_map(h->n);
                hipLaunchKernelGGL(mobile_sample_actions_k, dim3((unsigned)(plane_rows(p, T) * pm.bpr)), dim3(kBlock), 0, h->stream,
                                   p, h->rng, T, pm, h->act_plane[0]);
                SRL_HIP_CHECK(h, hipGetLastError());
            }
            int rc = launch_rollout_ep(h, p, T, h->act_plane[cur], d_obs, d_rew, d_done, 1, h->act_plane[cur ^ 1], d_act_out);
            if (rc) return rc;
            SRL_HIP_CHECK(h, hipGetLastError());
            h->prefetch_valid = true; h->prefetch_T = T; h->prefetch_buf = cur ^ 1;
            return 0;
        }
    }
    if (!d_actions) {
        h->prefetch_valid = false;              // this rollout moves the action-stream counters past what was drawn ahead
        // synthetic agent: draw the whole [T][N] action plane in parallel, into the caller's plane when there is one
        const size_t bytes = (size_t)T * h->n * (p.is_discrete ? 4 : 8);
        void *plane = d_act_out;
        if (!plane) {
            if (h->st_noise_sz < bytes) {
                if (h->st_noise) (void)hipFree(h->st_noise);
                h->st_noise = nullptr; h->st_noise_sz = 0;
                SRL_HIP_CHECK(h, hipMalloc(&h->st_noise, bytes));
                h->st_noise_sz = bytes;
            }
            plane = h->st_noise;
        }
        const PlaneMap pm = plane_map(h->n);
        hipLaunchKernelGGL(mobile_sample_actions_k, dim3((unsigned)(plane_rows(p, T) * pm.bpr)), dim3(kBlock), 0, h->stream,
                           p, h->rng, T, pm, plane);
        SRL_HIP_CHECK(h, hipGetLastError());
        d_actions = plane;
        advance = 1;
    }
    // counter-based streams + fixed-length episodes: segments of the rollout run in parallel (mobile_rollout_ep_k)
    if (ep_path) { int rc = launch_rollout_ep(h, p, T, d_actions, d_obs, d_rew, d_done, advance, nullptr, nullptr); if (rc) return rc; }
    else h->snap_valid = false;          // the sequential kernels below move the live state only
    if (ep_path) {}
    else if (h->cfg.rng_mode == SRLHIP_RNG_PHILOX) launch_rollout<SRLHIP_RNG_PHILOX>(h, p, T, d_actions, nullptr, d_obs, d_rew, d_done, advance);
    else launch_rollout<SRLHIP_RNG_MT19937>(h, p, T, d_actions, nullptr, d_obs, d_rew, d_done, advance);
    SRL_HIP_CHECK(h, hipGetLastError());
    return 0;
}

int mobile_step(Handle *h, const void *d_actions, const double *d_noise, float *d_obs, float *d_rew,
                uint8_t *d_done) {
    if (h->cfg.rng_mode != SRLHIP_RNG_HOST) return mobile_rollout(h, 1, d_actions, d_obs, d_rew, d_done, nullptr);
    h->snap_valid = false;
    MobileParams p = params_of(h);
    dim3 grid((h->n + kBlock - 1) / kBlock), block(kBlock);
    hipLaunchKernelGGL((mobile_rollout_k<SRLHIP_RNG_HOST, -1, -1>), grid, block, 0, h->stream, p, h->mobile, h->rng, h->stats,
                       1, d_actions, d_noise, d_obs, d_rew, d_done, 0, PersistArgs{});
    SRL_HIP_CHECK(h, hipGetLastError());
    return 0;
}

// ---- persistent stepping, host side (api.hip: srlhip_set_persistent) --------------------------------------------------------------------
template <int MODE> static const void *mobile_persist_fn(const MobileParams &p) {
#define SRL_FN(KIND, DISC) reinterpret_cast<const void *>(mobile_persist_k<MODE, KIND, DISC>)
#define SRL_KIND(KIND) (p.is_discrete ? SRL_FN(KIND, 1) : SRL_FN(KIND, 0))
    switch (p.kind) {
        case SRLHIP_ENV_MOBILE: return SRL_KIND(SRLHIP_ENV_MOBILE);
        case SRLHIP_ENV_MOBILE_1D: return SRL_KIND(SRLHIP_ENV_MOBILE_1D);
        case SRLHIP_ENV_MOBILE_2TARGET: return SRL_KIND(SRLHIP_ENV_MOBILE_2TARGET);
        default: return SRL_KIND(SRLHIP_ENV_MOBILE_LINE);
    }
#undef SRL_KIND
#undef SRL_FN
}
// workgroups of the resident kernel (0: no persistent form for this handle); capacity: how many the device holds at once;
// eighths: which of the 8 `done` words it writes (eighth g = the workgroups b = g mod 8)
int mobile_persist_blocks(Handle *h, int *capacity, uint32_t *eighths) {
    if (capacity) *capacity = 0;
    const srlhip_config &c = h->cfg;
    if (!(c.rng_mode == SRLHIP_RNG_PHILOX || c.rng_mode == SRLHIP_RNG_MT19937) || c.obs_mode == SRLHIP_OBS_RAW_PIXELS) return 0;
    const MobileParams p = params_of(h);
    const int grid = (h->n + kBlock - 1) / kBlock;
    int per_cu = 0;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, c.device_id) != hipSuccess) return 0;
    const void *fn = c.rng_mode == SRLHIP_RNG_PHILOX ? mobile_persist_fn<SRLHIP_RNG_PHILOX>(p) : mobile_persist_fn<SRLHIP_RNG_MT19937>(p);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, kBlock, 0) != hipSuccess) return 0;
    if (capacity) *capacity = per_cu * prop.multiProcessorCount;
    if (eighths) *eighths = grid >= 8 ? 0xffu : (1u << grid) - 1u;
    return (long long)per_cu * prop.multiProcessorCount >= grid ? grid : 0;
}
int mobile_persist_start(Handle *h, const void *d_actions, float *d_obs, float *d_rew, uint8_t *d_done, const PersistArgs &pa) {
    h->snap_valid = false; h->prefetch_valid = false;
    const MobileParams p = params_of(h);
    dim3 grid((h->n + kBlock - 1) / kBlock), block(kBlock);
#define SRL_GO(MODE, KIND, DISC) hipLaunchKernelGGL((mobile_persist_k<MODE, KIND, DISC>), grid, block, 0, h->stream, p, h->mobile, h->rng, h->stats, d_actions, d_obs, d_rew, d_done, pa)
#define SRL_KIND(MODE, KIND) { if (p.is_discrete) SRL_GO(MODE, KIND, 1); else SRL_GO(MODE, KIND, 0); }
#define SRL_MODE(MODE)                                                                        \
    switch (p.kind) {                                                                         \
        case SRLHIP_ENV_MOBILE: SRL_KIND(MODE, SRLHIP_ENV_MOBILE) break;                       \
        case SRLHIP_ENV_MOBILE_1D: SRL_KIND(MODE, SRLHIP_ENV_MOBILE_1D) break;                 \
        case SRLHIP_ENV_MOBILE_2TARGET: SRL_KIND(MODE, SRLHIP_ENV_MOBILE_2TARGET) break;       \
        default: SRL_KIND(MODE, SRLHIP_ENV_MOBILE_LINE)                                        \
    }
    if (h->cfg.rng_mode == SRLHIP_RNG_PHILOX) SRL_MODE(SRLHIP_RNG_PHILOX) else SRL_MODE(SRLHIP_RNG_MT19937)
#undef SRL_MODE
#undef SRL_KIND
#undef SRL_GO
    SRL_HIP_CHECK(h, hipGetLastError());
    return 0;
}

int mobile_field(Handle *h, int field, void **dptr, size_t *elem, int *count) {
    MobileState &s = h->mobile;
    *count = 1;
    switch (field) {
        case SRLHIP_F_POS_X: *dptr = s.pos_x; *elem = 8; return 0;
        case SRLHIP_F_POS_Y: *dptr = s.pos_y; *elem = 8; return 0;
        case SRLHIP_F_TARGET_X: *dptr = s.tgt_x; *elem = 8; return 0;
        case SRLHIP_F_TARGET_Y: *dptr = s.tgt_y; *elem = 8; return 0;
        case SRLHIP_F_TARGET2_X: *dptr = s.tgt2_x; *elem = 8; return 0;
        case SRLHIP_F_TARGET2_Y: *dptr = s.tgt2_y; *elem = 8; return 0;
        case SRLHIP_F_STEP_COUNT: *dptr = s.counter; *elem = 4; return 0;
        case SRLHIP_F_CUR_TARGET: *dptr = s.cur_target; *elem = 4; return 0;
    }
    return h->fail(SRLHIP_EINVAL, "unknown field for the MobileRobot family");
}

}  // namespace srl
