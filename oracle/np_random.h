/* np_random.h — plain-C restatement of numpy's legacy RandomState (MT19937)
 * as used behind the reference's `self.np_random`
 * (/root/reference/environments/srl_env.py:71-78; numpy pinned 1.14.0,
 * /root/reference/environment.yml:47 — the legacy stream is frozen).
 * Published algorithm: Matsumoto & Nishimura mt19937ar.c + numpy randomkit
 * rk_double / rk_gauss / rk_random_uint32 (masked rejection).
 * TEST INFRASTRUCTURE ONLY.  Checked against numpy in tests/test_oracle_c.py. */
#ifndef ORACLE_NP_RANDOM_H
#define ORACLE_NP_RANDOM_H
#include <math.h>
#include <stdint.h>

typedef struct {
    uint32_t mt[624];
    int pos;
    int has_gauss;
    double gauss;
} np_rng;

static void np_rng_seed_array(np_rng *r, const uint32_t *key, int key_len) {
    int i, j, k;
    r->mt[0] = 19650218u;
    for (i = 1; i < 624; i++) r->mt[i] = 1812433253u * (r->mt[i - 1] ^ (r->mt[i - 1] >> 30)) + (uint32_t)i;
    i = 1; j = 0;
    k = 624 > key_len ? 624 : key_len;
    for (; k; k--) {
        r->mt[i] = (r->mt[i] ^ ((r->mt[i - 1] ^ (r->mt[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
        i++; j++;
        if (i >= 624) { r->mt[0] = r->mt[623]; i = 1; }
        if (j >= key_len) j = 0;
    }
    for (k = 623; k; k--) {
        r->mt[i] = (r->mt[i] ^ ((r->mt[i - 1] ^ (r->mt[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
        i++;
        if (i >= 624) { r->mt[0] = r->mt[623]; i = 1; }
    }
    r->mt[0] = 0x80000000u;
    r->pos = 624; r->has_gauss = 0; r->gauss = 0.0;
}

static uint32_t np_rng_u32(np_rng *r) {
    uint32_t y;
    if (r->pos == 624) {
        int i;
        for (i = 0; i < 624 - 397; i++) {
            y = (r->mt[i] & 0x80000000u) | (r->mt[i + 1] & 0x7fffffffu);
            r->mt[i] = r->mt[i + 397] ^ (y >> 1) ^ (-(int32_t)(y & 1) & 0x9908b0dfu);
        }
        for (; i < 623; i++) {
            y = (r->mt[i] & 0x80000000u) | (r->mt[i + 1] & 0x7fffffffu);
            r->mt[i] = r->mt[i + (397 - 624)] ^ (y >> 1) ^ (-(int32_t)(y & 1) & 0x9908b0dfu);
        }
        y = (r->mt[623] & 0x80000000u) | (r->mt[0] & 0x7fffffffu);
        r->mt[623] = r->mt[396] ^ (y >> 1) ^ (-(int32_t)(y & 1) & 0x9908b0dfu);
        r->pos = 0;
    }
    y = r->mt[r->pos++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

static double np_rng_double(np_rng *r) {
    long a = np_rng_u32(r) >> 5, b = np_rng_u32(r) >> 6;
    return (a * 67108864.0 + b) / 9007199254740992.0;
}

static double np_rng_gauss(np_rng *r) {
    if (r->has_gauss) {
        const double tmp = r->gauss;
        r->gauss = 0; r->has_gauss = 0;
        return tmp;
    } else {
        double f, x1, x2, r2;
        do {
            x1 = 2.0 * np_rng_double(r) - 1.0;
            x2 = 2.0 * np_rng_double(r) - 1.0;
            r2 = x1 * x1 + x2 * x2;
        } while (r2 >= 1.0 || r2 == 0.0);
        f = sqrt(-2.0 * log(r2) / r2);
        r->gauss = f * x1; r->has_gauss = 1;
        return f * x2;
    }
}

static double np_rng_normal(np_rng *r, double loc, double scale) { return loc + scale * np_rng_gauss(r); }
static double np_rng_uniform(np_rng *r, double low, double high) { return low + (high - low) * np_rng_double(r); }
/* RandomState.randint(n) for 0 < n <= 2^32: masked rejection on 32-bit draws */
static uint32_t np_rng_randint(np_rng *r, uint32_t n) {
    uint32_t rng = n - 1, mask = rng, v;
    if (rng == 0) return 0;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
    while ((v = (np_rng_u32(r) & mask)) > rng) {}
    return v;
}
#endif
