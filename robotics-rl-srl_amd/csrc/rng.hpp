// rng.hpp — random streams of the env stepper (host + gfx950 device code).
//
//  * Mt19937View : numpy `RandomState` (legacy MT19937) resident in HBM, one
//    generator per env, stored structure-of-arrays (word k of env e at
//    mt[k*stride + e]) so a wavefront's 64 envs read one coalesced row.
//    Reproduces the stream behind the reference's `self.np_random`
//    (environments/srl_env.py:71-78 -> gym.utils.seeding.np_random ->
//    numpy.random.RandomState): random_sample, uniform, normal (polar method
//    with the cached second deviate), randint (masked rejection).
//  * gym_hash_seed() : gym==0.11.0 seeding.hash_seed (sha512 of str(seed)).
//  * Philox4x32-10 : counter-based stream for the throughput mode.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SRL_HD __host__ __device__ __forceinline__

namespace srl {

// ------------------------------------------------------------------ MT19937
constexpr int MT_N = 624;
constexpr int MT_M = 397;

// numpy's derived draws on top of any source of tempered 32-bit words (G::u32()): shared by the per-env generator below and by
// the lane-group generator of the Kuka kernels (kuka_group.hpp: GroupMt), so that both produce the same stream bit for bit.
// numpy rk_double / random_sample: 53-bit double in [0, 1)
template <class G> SRL_HD double mt_double01(G &gen) {
    uint32_t a = gen.u32() >> 5, b = gen.u32() >> 6;
    return ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
}
// legacy_gauss (polar Box-Muller, caches the second deviate)
template <class G> SRL_HD double mt_std_normal(G &gen, int32_t &has_g, double &g) {
#pragma clang fp contract(off)
    if (has_g) { double t = g; g = 0.0; has_g = 0; return t; }
    double x1, x2, r2;
    do {
        x1 = 2.0 * mt_double01(gen) - 1.0;
        x2 = 2.0 * mt_double01(gen) - 1.0;
        r2 = x1 * x1 + x2 * x2;
    } while (r2 >= 1.0 || r2 == 0.0);
    double f = sqrt(-2.0 * log(r2) / r2);
    g = f * x1; has_g = 1;
    return f * x2;
}
template <class G> SRL_HD double mt_normal(G &gen, int32_t &has_g, double &g, double loc, double scale) {
#pragma clang fp contract(off)   // numpy computes these unfused; keep them bit-identical under -ffp-contract=fast
    return loc + scale * mt_std_normal(gen, has_g, g);
}
template <class G> SRL_HD double mt_uniform(G &gen, double low, double high) {
#pragma clang fp contract(off)   // numpy computes these unfused; keep them bit-identical under -ffp-contract=fast
    return low + (high - low) * mt_double01(gen);
}
// RandomState.randint(0, rng+1): masked rejection on 32-bit draws (rng < 2^32)
template <class G> SRL_HD uint32_t mt_bounded(G &gen, uint32_t rng) {
    if (rng == 0) return 0;
    uint32_t mask = rng;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
    uint32_t v;
    while ((v = (gen.u32() & mask)) > rng) {}
    return v;
}
SRL_HD uint32_t mt_temper(uint32_t y) {
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

struct Mt19937View {
    uint32_t *mt;        // [MT_N][stride]
    int32_t *mti;        // [stride]
    int32_t *has_gauss;  // [stride]
    double *gauss;       // [stride]
    int64_t stride;
};

// One generator addressed through the SoA view; keeps the index in a register.
struct Mt19937 {
    uint32_t *w;   // &mt[env]
    int64_t s;     // stride
    int32_t idx;
    int32_t has_g;
    double g;

    SRL_HD void load(const Mt19937View &v, int64_t env) {
        w = v.mt + env; s = v.stride; idx = v.mti[env]; has_g = v.has_gauss[env]; g = v.gauss[env];
    }
    SRL_HD void store(const Mt19937View &v, int64_t env) const {
        v.mti[env] = idx; v.has_gauss[env] = has_g; v.gauss[env] = g;
    }
    SRL_HD uint32_t &at(int k) { return w[(int64_t)k * s]; }

    // init_genrand + init_by_array, as numpy's RandomState.seed(array) does.
    SRL_HD void seed_by_array(const uint32_t *key, int key_len) {
        at(0) = 19650218u;
        for (int i = 1; i < MT_N; i++) {
            uint32_t p = at(i - 1);
            at(i) = 1812433253u * (p ^ (p >> 30)) + (uint32_t)i;
        }
        int i = 1, j = 0;
        int k = MT_N > key_len ? MT_N : key_len;
        for (; k; k--) {
            uint32_t p = at(i - 1);
            at(i) = (at(i) ^ ((p ^ (p >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
            i++; j++;
            if (i >= MT_N) { at(0) = at(MT_N - 1); i = 1; }
            if (j >= key_len) j = 0;
        }
        for (k = MT_N - 1; k; k--) {
            uint32_t p = at(i - 1);
            at(i) = (at(i) ^ ((p ^ (p >> 30)) * 1566083941u)) - (uint32_t)i;
            i++;
            if (i >= MT_N) { at(0) = at(MT_N - 1); i = 1; }
        }
        at(0) = 0x80000000u;
        idx = MT_N; has_g = 0; g = 0.0;
    }

    // Regenerates the 624 words in place.  Word k needs the OLD words k, k+1 and word (k+397) mod 624 — old for k < 227,
    // already regenerated for k >= 227.  Done in blocks of 16 (624 = 39 x 16): all 33 reads of a block are issued before
    // its 16 writes, so the memory system sees batches instead of 624 dependent load/store pairs (the state lives in
    // HBM; a wavefront waits for its slowest lane: the per-word loop cost ~0.6 ms per twist, this ~10x less).  A block
    // never reads a word it writes itself (the offsets are 1 — read before the writes — and 397 / 227), so the result
    // is the sequential algorithm's, bit for bit.
    __host__ __device__ __attribute__((noinline)) void twist() {      // rare and large: kept out of line
        constexpr int B = 16;
        for (int b = 0; b < MT_N; b += B) {
            uint32_t a[B + 1], m[B];
            const int bm = b + MT_M >= MT_N ? b + MT_M - MT_N : b + MT_M;     // (b + 397) mod 624
#pragma unroll
            for (int i = 0; i <= B; i++) { const int k = b + i; a[i] = at(k >= MT_N ? k - MT_N : k); }
#pragma unroll
            for (int i = 0; i < B; i++) { const int k = bm + i; m[i] = at(k >= MT_N ? k - MT_N : k); }
#pragma unroll
            for (int i = 0; i < B; i++) {
                const uint32_t y = (a[i] & 0x80000000u) | (a[i + 1] & 0x7fffffffu);
                uint32_t v = m[i] ^ (y >> 1);
                if (y & 1u) v ^= 0x9908b0dfu;
                at(b + i) = v;
            }
        }
        idx = 0;
    }

    SRL_HD uint32_t u32() {
        if (idx >= MT_N) twist();
        return mt_temper(at(idx++));
    }
    SRL_HD double double01() { return mt_double01(*this); }
    SRL_HD double std_normal() { return mt_std_normal(*this, has_g, g); }
    SRL_HD double normal(double loc, double scale) { return mt_normal(*this, has_g, g, loc, scale); }
    SRL_HD double uniform(double low, double high) { return mt_uniform(*this, low, high); }
    SRL_HD uint32_t bounded(uint32_t rng) { return mt_bounded(*this, rng); }
};

// ------------------------------------------------------------- Philox4x32-10
struct Philox {
    uint32_t k0, k1;   // key   = seed of the env (lo, hi)
    uint64_t ctr;      // block counter (per env, persisted)
    uint32_t stream;   // 0 = env noise/reset draws, 1 = synthetic agent actions

    SRL_HD static void round(uint32_t &c0, uint32_t &c1, uint32_t &c2, uint32_t &c3, uint32_t a, uint32_t b) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ a;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ b;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    }
    SRL_HD void block(uint32_t out[4]) {
        uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32), c2 = stream, c3 = 0x5eed5eedu;
        uint32_t a = k0, b = k1;
#pragma unroll
        for (int r = 0; r < 10; r++) {
            round(c0, c1, c2, c3, a, b);
            a += 0x9E3779B9u; b += 0xBB67AE85u;
        }
        out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
        ctr++;
    }
    SRL_HD static double to_double(uint32_t a, uint32_t b) {
        return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) / 9007199254740992.0;
    }
    SRL_HD double double01() { uint32_t o[4]; block(o); return to_double(o[0], o[1]); }
    SRL_HD double uniform(double low, double high) {
#pragma clang fp contract(off)   // numpy computes these unfused; keep them bit-identical under -ffp-contract=fast
        return low + (high - low) * double01();
    }
    // Box-Muller on one block, no caching (counter-based streams stay random-access)
    SRL_HD double std_normal() {
#pragma clang fp contract(off)
        uint32_t o[4]; block(o);
        double u1 = 1.0 - to_double(o[0], o[1]);   // (0, 1]
        double u2 = to_double(o[2], o[3]);
        return sqrt(-2.0 * log(u1)) * cos(6.283185307179586476925286766559 * u2);
    }
    SRL_HD double normal(double loc, double scale) {
#pragma clang fp contract(off)   // numpy computes these unfused; keep them bit-identical under -ffp-contract=fast
        return loc + scale * std_normal();
    }
    SRL_HD uint32_t bounded(uint32_t rng) {       // multiply-shift into [0, rng]
        uint32_t o[4]; block(o);
        return (uint32_t)(((uint64_t)o[0] * ((uint64_t)rng + 1)) >> 32);
    }
};

// ------------------------------------------------------------------- sha512
// Host only: gym.utils.seeding.hash_seed -> up to two uint32 digits.
int gym_hash_seed(uint64_t seed, uint32_t digits[2]);

}  // namespace srl
