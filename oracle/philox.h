/* philox.h — Philox4x32-10 (Salmon et al., SC'11) as laid out by the stepper's
 * throughput mode (SRLHIP_RNG_PHILOX): key = env seed (lo, hi), counter =
 * (block lo, block hi, stream, 0x5eed5eed).  This is this repo's own synthetic
 * input generator (no reference counterpart); the oracle restates it so that
 * device-generated random-agent rollouts can be replayed on the CPU.
 * TEST INFRASTRUCTURE ONLY. */
#ifndef ORACLE_PHILOX_H
#define ORACLE_PHILOX_H
#include <math.h>
#include <stdint.h>

typedef struct { uint32_t k0, k1; uint64_t ctr; uint32_t stream; } philox_t;

static void philox_block(philox_t *p, uint32_t out[4]) {
    uint32_t c0 = (uint32_t)p->ctr, c1 = (uint32_t)(p->ctr >> 32), c2 = p->stream, c3 = 0x5eed5eedu;
    uint32_t a = p->k0, b = p->k1;
    int r;
    for (r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ a, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ b, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        a += 0x9E3779B9u; b += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
    p->ctr++;
}
static double philox_to_double(uint32_t a, uint32_t b) {
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) / 9007199254740992.0;
}
static double philox_double(philox_t *p) { uint32_t o[4]; philox_block(p, o); return philox_to_double(o[0], o[1]); }
static double philox_uniform(philox_t *p, double lo, double hi) { return lo + (hi - lo) * philox_double(p); }
static double philox_std_normal(philox_t *p) {
    uint32_t o[4]; double u1, u2;
    philox_block(p, o);
    u1 = 1.0 - philox_to_double(o[0], o[1]);
    u2 = philox_to_double(o[2], o[3]);
    return sqrt(-2.0 * log(u1)) * cos(6.283185307179586476925286766559 * u2);
}
static uint32_t philox_bounded(philox_t *p, uint32_t rng) {
    uint32_t o[4]; philox_block(p, o);
    return (uint32_t)(((uint64_t)o[0] * ((uint64_t)rng + 1)) >> 32);
}
#endif
