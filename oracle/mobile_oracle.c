/* mobile_oracle.c — plain-C, scalar float64 restatement of the MobileRobot env
 * family for bulk parity runs (N = 4096 envs).  TEST INFRASTRUCTURE ONLY: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * Follows /root/reference/environments/mobile_robot/
 *   mobile_robot_env.py        reset :159-181, step :235-280, _termination :336-343, _reward :345-363
 *   mobile_robot_1D_env.py     reset :58-74,   step :108-147
 *   mobile_robot_2target_env.py reset :35-67,  _reward :162-181
 *   mobile_robot_line_target_env.py :3-4, :35-40, :56-64, :108-125
 * and is itself pinned against oracle/mobile_oracle.py / tests/golden/mobile_reference.npz
 * (vectors produced by the reference source) in tests/test_oracle_c.py.
 * Compile with -ffp-contract=off: the only fused operations are the explicit
 * fma() calls restating BLAS ddot inside np.linalg.norm.                     */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "np_random.h"
#include "philox.h"

enum { MOBILE = 0, MOBILE_1D = 1, MOBILE_2TARGET = 2, MOBILE_LINE = 3 };
enum { RNG_PHILOX = 1, RNG_MT19937 = 2 };

typedef struct {
    double x, y, tx, ty, t2x, t2y;
    int counter, cur;
} menv;

typedef struct {
    int mode;
    np_rng mt;
    philox_t ph;
} mrng;

static double r_uniform(mrng *r, double lo, double hi) {
    return r->mode == RNG_MT19937 ? np_rng_uniform(&r->mt, lo, hi) : philox_uniform(&r->ph, lo, hi);
}
static double r_normal(mrng *r, double loc, double scale) {
    if (r->mode == RNG_MT19937) return np_rng_normal(&r->mt, loc, scale);
    return scale == 0.0 ? loc : loc + scale * philox_std_normal(&r->ph);
}

static void m_reset(menv *m, mrng *r, int kind, int random_target) {
    const double max_x = 4.0, max_y = 4.0, margin = 0.1 * 4.0;
    m->cur = 0;
    m->x = max_x / 2 + r_uniform(r, -max_x / 3, max_x / 3);
    m->y = 0.0;
    if (kind != MOBILE_1D) m->y = max_y / 2 + r_uniform(r, -max_y / 3, max_y / 3);
    m->tx = 0.9 * max_x; m->ty = 0.0; m->t2x = 0.0; m->t2y = 0.0;
    if (kind == MOBILE_1D) {
        if (random_target) m->tx = r_uniform(r, 0 + margin, max_x - margin);
    } else if (kind == MOBILE_LINE) {
        if (random_target) m->tx = r_uniform(r, 0 + margin, max_x - margin);
        m->ty = max_x;
    } else {
        m->ty = max_y * 3 / 4;
        if (random_target) {
            m->tx = r_uniform(r, 0 + margin, max_x - margin);
            m->ty = r_uniform(r, 0 + margin, max_y - margin);
        }
        if (kind == MOBILE_2TARGET) {
            m->t2x = 0.1 * max_x; m->t2y = max_y * 3 / 4;
            if (random_target) {
                m->t2x = r_uniform(r, 0 + margin, max_x - margin);
                m->t2y = r_uniform(r, 0 + margin, max_y - margin);
            }
        }
    }
    m->counter = 0;
}

static void m_obs(const menv *m, int kind, float *o) {
    double tx = m->cur ? m->t2x : m->tx, ty = m->cur ? m->t2y : m->ty;
    if (kind == MOBILE_LINE) {
        double t = tx - 0.2;
        o[0] = (float)(m->x - t); o[1] = (float)(m->y - t);
    } else if (kind == MOBILE_1D) {
        o[0] = (float)(m->x - tx);
    } else {
        o[0] = (float)(m->x - tx); o[1] = (float)(m->y - ty);
    }
}

static double m_step(menv *m, int kind, int is_discrete, int shape_reward, int a, const float *af, double dv,
                     int *done) {
    double dx = 0.0, dy = 0.0, px = m->x, py = m->y, distance, threshold = 0.4, reward = 0.0, tx, ty;
    int bumped = 0;
    const double robot_dim[2] = {0.325 * 2, 0.2};
    if (is_discrete) {
        if (kind == MOBILE_1D) { if (a == 0) dx = -dv; else if (a == 1) dx = dv; }
        else {
            if (a == 0) dx = -dv; else if (a == 1) dx = dv;
            else if (a == 2) dy = -dv; else if (a == 3) dy = dv;
        }
    } else {
        float fdv = (float)dv, c0 = fmaxf(fminf(af[0], 1.0f), -1.0f), c1 = fmaxf(fminf(af[1], 1.0f), -1.0f);
        dx = (double)(c0 * fdv); dy = (double)(c1 * fdv);
    }
    m->x = m->x + dx;
    if (kind != MOBILE_1D) m->y = m->y + dy;
    {
        int naxis = kind == MOBILE_1D ? 1 : 2, i;
        for (i = 0; i < naxis; i++) {
            double margin = 0.1 + robot_dim[i] / 2, v = i == 0 ? m->x : m->y;
            if (v < margin || v > 4 - margin) { bumped = 1; m->x = px; m->y = py; break; }
        }
    }
    m->counter += 1;
    tx = m->cur ? m->t2x : m->tx; ty = m->cur ? m->t2y : m->ty;
    if (kind == MOBILE_LINE) { distance = fabs((tx - 0.2) - m->x); threshold = 0.1; }
    else if (kind == MOBILE_1D) { double d0 = tx - m->x; distance = sqrt(fma(d0, d0, 0.0)); }
    else { double d0 = tx - m->x, d1 = ty - m->y; distance = sqrt(fma(d1, d1, fma(d0, d0, 0.0))); }
    if (distance <= threshold) {
        reward = 1.0;
        if (kind == MOBILE_2TARGET && m->cur < 1) m->cur += 1;
    }
    if (bumped) reward = -1.0;
    if (shape_reward) reward = -distance;
    *done = m->counter > 250;
    return reward;
}

/* Auto-resetting rollout of n independent envs for T steps after an initial reset
 * (SB VecEnv worker semantics).  actions == NULL -> synthetic random agent drawn
 * from the Philox action stream (stream 1), as the device does.
 * Layouts: obs0 [n][od], obs [T][n][od], rew/rew64/done [T][n], final_state [n][8]
 * (x, y, tx, ty, t2x, t2y, counter, cur), ep_stats [n][3] (last_return, last_length, n_finished). */
int mobile_oracle_rollout(int kind, int is_discrete, int random_target, int shape_reward, int rng_mode, int n, int T,
                          const int64_t *seeds, const uint32_t *mt_keys, const int32_t *mt_key_len,
                          const void *actions, float *obs0, float *obs, float *rew, double *rew64, uint8_t *done_out,
                          void *act_out, double *final_state, double *ep_stats) {
    const int od = kind == MOBILE_1D ? 1 : 2;
    int e, t;
    mrng *r = (mrng *)malloc(sizeof(mrng));
    if (!r) return -12;
    for (e = 0; e < n; e++) {
        menv m; philox_t act; uint64_t act_i = 0;
        double ep_ret = 0.0, last_ret = 0.0; int ep_len = 0, last_len = 0, n_fin = 0;
        memset(&m, 0, sizeof m);
        r->mode = rng_mode;
        if (rng_mode == RNG_MT19937) np_rng_seed_array(&r->mt, mt_keys + 2 * (size_t)e, mt_key_len[e]);
        r->ph.k0 = (uint32_t)(uint64_t)seeds[e]; r->ph.k1 = (uint32_t)((uint64_t)seeds[e] >> 32);
        r->ph.ctr = 0; r->ph.stream = 0;
        act = r->ph; act.stream = 1;
        m_reset(&m, r, kind, random_target);
        if (obs0) m_obs(&m, kind, obs0 + (size_t)e * od);
        for (t = 0; t < T; t++) {
            size_t row = (size_t)t * n + e;
            int a = 0, done; float af[2] = {0.f, 0.f}; double dv, reward;
            if (actions) {
                if (is_discrete) a = ((const int32_t *)actions)[row];
                else { af[0] = ((const float *)actions)[2 * row]; af[1] = ((const float *)actions)[2 * row + 1]; }
            } else {
                if (is_discrete) {
                    /* synthetic agent, discrete: action i of the env is word i % 4 of block i / 4 of its action stream (multiply-shift
                     * into [0, m]) — four actions per Philox block (round 6; before: one block per action, word 0) */
                    uint32_t o[4]; const uint32_t m = kind == MOBILE_1D ? 1u : 3u;
                    act.ctr = act_i >> 2; philox_block(&act, o);
                    a = (int)(uint32_t)(((uint64_t)o[act_i & 3] * ((uint64_t)m + 1)) >> 32);
                    act_i++;
                }
                else {
                    uint32_t o[4]; philox_block(&act, o);
                    af[0] = (float)(-1.0 + 2.0 * philox_to_double(o[0], o[1]));
                    af[1] = (float)(-1.0 + 2.0 * philox_to_double(o[2], o[3]));
                }
                if (act_out) {
                    if (is_discrete) ((int32_t *)act_out)[row] = a;
                    else { ((float *)act_out)[2 * row] = af[0]; ((float *)act_out)[2 * row + 1] = af[1]; }
                }
            }
            dv = 0.1 + r_normal(r, 0.0, 0.0);
            reward = m_step(&m, kind, is_discrete, shape_reward, a, af, dv, &done);
            ep_ret += reward; ep_len += 1;
            if (done) {
                last_ret = ep_ret; last_len = ep_len; n_fin += 1; ep_ret = 0.0; ep_len = 0;
                m_reset(&m, r, kind, random_target);
            }
            if (obs) m_obs(&m, kind, obs + row * od);
            if (rew) rew[row] = (float)reward;
            if (rew64) rew64[row] = reward;
            if (done_out) done_out[row] = (uint8_t)done;
        }
        if (final_state) {
            double *f = final_state + 8 * (size_t)e;
            f[0] = m.x; f[1] = m.y; f[2] = m.tx; f[3] = m.ty; f[4] = m.t2x; f[5] = m.t2y; f[6] = m.counter; f[7] = m.cur;
        }
        if (ep_stats) { ep_stats[3 * (size_t)e] = last_ret; ep_stats[3 * (size_t)e + 1] = last_len; ep_stats[3 * (size_t)e + 2] = n_fin; }
    }
    free(r);
    return 0;
}

/* numpy-stream self test hook: draws from a RandomState seeded with key. */
void oracle_np_random_draws(const uint32_t *key, int key_len, int n, double *uniform01, double *gauss,
                            uint32_t *randint3) {
    np_rng r; int i;
    np_rng_seed_array(&r, key, key_len);
    for (i = 0; i < n; i++) uniform01[i] = np_rng_double(&r);
    for (i = 0; i < n; i++) gauss[i] = np_rng_gauss(&r);
    for (i = 0; i < n; i++) randint3[i] = np_rng_randint(&r, 3);
}
