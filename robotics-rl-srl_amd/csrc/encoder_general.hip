// encoder_general.hip — layered SRL encoder forward (srl_zoo CustomCNN) for any frame shape, gfx950.
//
// The fused kernel of encoder.hip keeps a whole 64x64x3 frame in one CU's LDS.  The reference's real observation is
// 224x224x3 (kuka_button_gym_env.py:21-22) or 6 channels with multi_view (:401-417): those maps do not fit a CU, so
// this path runs the network layer by layer — same arithmetic (implicit GEMM on v_mfma_f32_32x32x16_f16 with every
// operand split into f16 hi + lo so that the products accumulate to float32 accuracy, power-of-two weight pre-scale,
// ImageNet normalisation and zero padding folded into layer 1 through a validity-mask channel), different tiling:
//   * activations travel between layers as f16 hi / lo planes in HBM, [frame][row][column][64 channels];
//   * one 128-lane workgroup (two wavefronts = the two halves of the 64 output channels) owns a band of R pooled rows
//     by 15 pooled columns of one frame: it stages the input window it needs in LDS (rows and columns with their halo,
//     zero outside the map), each wavefront accumulates the 2R+1 convolution rows x 32 columns of its tile in
//     registers, and the 3x3/2 max-pool runs on those raw accumulators — vertically element-wise across the row
//     tiles, horizontally after one lane^32 exchange — so a convolution output never touches memory; only the pooled
//     quarter is rescaled, biased, rectified, split and stored;
//   * layer-1 weights (B fragments) stay on the CU while a workgroup walks several bands — in LDS, shared by the four
//     wavefronts of a two-segment workgroup (3-channel frames: two workgroups = eight wavefronts per CU), or in registers
//     (6-channel frames: twice the weights); layers 2-3 stream theirs from L2 through a register ring;
//   * the fully connected layer is its own small kernel over the [frame][Hp3][Wp3][64] float32 features.
// The network sees the frame with H and W swapped (models.py:185-188); all layers are symmetric in the two spatial
// dims, so the kernels work on the frame as rasterised and the packers swap the kernel axes / the FC's flatten order.
#include "encoder_general.hpp"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <new>
#include <vector>

#include "../../include/srlhip.h"

#ifndef EG_X
#define EG_X 0          // experiment builds (profiles/probes/encoder_general_experiments.sh): 1 = layers 2-3 without their MFMAs, 2 = without
#endif                  // the pooling / band stores, 3 = without the window loads; 4 / 5 / 6 = layer 1 without its MFMAs / pooling and
                        // stores / pixel unpack.  Results are wrong by construction; 0 = the product.

namespace srlenc {

Geometry geometry(int img_h, int img_w, int n_channels) {
    Geometry g;
    g.H = img_h; g.W = img_w; g.C = n_channels;
    int h = img_h, w = img_w;
    const int k[3] = {7, 3, 3}, s[3] = {2, 1, 2}, pad[3] = {3, 1, 1}, ppad[3] = {1, 0, 0};
    g.ok = (n_channels == 3 || n_channels == 6) && img_h >= 8 && img_w >= 8 && img_h <= 1024 && img_w <= 1024;
    for (int l = 0; l < 3; l++) {
        g.Hc[l] = (h + 2 * pad[l] - k[l]) / s[l] + 1; g.Wc[l] = (w + 2 * pad[l] - k[l]) / s[l] + 1;
        g.Hp[l] = g.Hc[l] + 2 * ppad[l] >= 3 ? (g.Hc[l] + 2 * ppad[l] - 3) / 2 + 1 : 0;
        g.Wp[l] = g.Wc[l] + 2 * ppad[l] >= 3 ? (g.Wc[l] + 2 * ppad[l] - 3) / 2 + 1 : 0;
        if (g.Hp[l] < 1 || g.Wp[l] < 1) g.ok = false;
        h = g.Hp[l]; w = g.Wp[l];
    }
    return g;
}

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4v __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

constexpr int PX = 144;                 // LDS pixel pitch of a 64-channel f16 plane (128 B + 16 B skew)
constexpr float kF16Max = 65504.f, kNegInf = -3.0e38f;
constexpr float kMean[3] = {0.485f, 0.456f, 0.406f}, kStd[3] = {0.229f, 0.224f, 0.225f};
constexpr int kNC1 = 70, kNC2 = 34, kNC3 = 65;          // staged input columns of a 32-column convolution tile
constexpr int kNC3Narrow = 33;                           // layer 3 when its map has at most 16 columns (224x224 frames: 14)
constexpr int kLdsSlack3 = 2 * 32 * PX;                  // the masked columns of a narrow tile still read (finite garbage) behind the window
constexpr int kR1 = 2, kR1x6 = 2, kR2 = 2, kR3 = 1;      // pooled rows per band

struct LayerParams {
    const uint8_t *img;              // layer 1: [n][H][W][C]
    const _Float16 *in_hi, *in_lo;   // layers 2, 3: [n][Hin][Win][64]
    _Float16 *out_hi, *out_lo;       // layers 1, 2: [n][Hp][Wp][64]
    float *out_f32;                  // layer 3
    const char *pack;                // B fragments [channel half][k-step][lane][8 hi | 8 lo] f16
    const float *bias;               // [64], null for layer 1 (its bias rides on the mask channel)
    float inv_scale;
    const float *inv1c;              // int8 layer 1: [64] per-output-channel 256 / fixed-point scale
    int Hin, Win, Cimg, Hc, Wc, Hp, Wp, nbands, bands_per_wg;
    int *status;
};

extern __shared__ __attribute__((aligned(16))) char lds[];

__device__ __forceinline__ half8 lds16(int off) { return *reinterpret_cast<const half8 *>(lds + off); }
__device__ __forceinline__ half8 glb16(const char *p) { return *reinterpret_cast<const half8 *>(p); }
__device__ __forceinline__ f32x16 mfma16(half8 a, half8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ float max3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

// LAYER 1: u8 frame -> conv7x7/2 p3 -> pool3/2 p1 (CPIX f16 per staged pixel: 4 = RGB + mask, 8 = 6 channels + mask + 0)
// LAYER 2: planes -> conv3x3 p1 -> pool3/2      LAYER 3: planes -> conv3x3/2 p1 -> pool3/2 -> float32 features
// NC: staged input columns (kNC1 / kNC2 / kNC3, or kNC3Narrow when layer 3's map is at most 16 columns wide)
// LDSB (layer 1, 3 channels): four wavefronts per workgroup — two column segments x two channel halves — with the layer's
// weights in LDS (56 KiB, shared) instead of 112 registers per lane: under 256 registers two workgroups = eight wavefronts
// fit a CU, and one wavefront's MFMAs overlap another's staging / pooling.
// I8 (layer 1, 3-channel frames, LDSB): the layer on the INT8 matrix pipe, exactly, as in the fused kernel (csrc/encoder.hip): the
// window holds the frame's bytes as (p - 128, mask 127) i8 pixels — 4 bytes per pixel, no f16 unpack —, the weights are three balanced
// base-256 digits of per-channel 24-bit fixed-point numbers (43 KiB of LDS instead of 56), one v_mfma_i32_32x32x32_i8 per kernel row
// and digit (105 MFMAs per band and wavefront where the f16 form takes 140), two convolution rows at a time with their three digit
// chains; the sums are recombined as 256 a0 + a1 + (a2 >> 8) in int32 (exact but for the last floor), then the float max-pool of the
// other layers.
template <int LAYER, int CPIX, int R, int NC, bool LDSB = false, bool I8 = false>
__global__ __launch_bounds__(LDSB ? 256 : 128, LDSB ? 2 : 1) void enc_layer_k(LayerParams P) {
    static_assert(!I8 || (LAYER == 1 && CPIX == 4 && LDSB), "the int8 form exists for layer 1 of 3-channel frames");
    constexpr int T = 2 * R + 1;                              // convolution rows of a band
    constexpr int KS = LAYER == 1 ? (I8 ? kI8Steps : CPIX == 4 ? 14 : 28) : 36;
    constexpr int S = LAYER == 2 ? 1 : 2, PAD = LAYER == 1 ? 3 : 1, KW = LAYER == 1 ? 7 : 3, PPAD = LAYER == 1 ? 1 : 0;
    constexpr int NR = S * (T - 1) + KW;                      // staged input rows
    constexpr int PIXB = LAYER == 1 ? (I8 ? 4 : CPIX * 2) : PX;   // staged bytes per pixel (and plane)
    constexpr int PLANE = NR * NC * PIXB;
    // tid: thread within its wavefront PAIR (the unit that owns a window); sp: the pair (column segment) inside an LDSB workgroup
    const int tid = threadIdx.x & 127, lane = tid & 63, nh = tid >> 6, m = lane & 31, h = lane >> 5, sp = threadIdx.x >> 7;
    const int seg = LDSB ? 2 * blockIdx.x + sp : blockIdx.x, img = blockIdx.z;
    constexpr int KSB = LAYER == 1 ? (CPIX == 4 ? 14 : 28) : 36;
    // LDSB: the layer's weights at the start of LDS — [channel half][k-step][lane][8 hi | 8 lo] f16, or [channel half][kernel row][digit][lane][16 i8]
    constexpr int WGT_BYTES = LDSB ? (I8 ? (int)kPackI8Bytes : 2 * KSB * 2048) : 0;
    constexpr int T_ = 2 * R + 1, NR_ = (LAYER == 2 ? 1 : 2) * (T_ - 1) + (LAYER == 1 ? 7 : 3);
    constexpr int WIN_BYTES = LDSB ? (NR_ * NC * PIXB > R * 15 * 256 ? NR_ * NC * PIXB : R * 15 * 256) : 0;
    char *const win = lds + WGT_BYTES + sp * WIN_BYTES;        // this pair's window, later its pooled band
    const int woff = WGT_BYTES + sp * WIN_BYTES;
    const int c0 = 30 * seg - PPAD, ix0 = S * c0 - PAD;
    const char *bp = P.pack + ((size_t)nh * KS * 64 + lane) * 32;
    half8 Bh1[LAYER == 1 && !LDSB ? KS : 1], Bl1[LAYER == 1 && !LDSB ? KS : 1];
    if constexpr (LDSB) {
        for (int c = threadIdx.x; c < WGT_BYTES / 16; c += 256)
            *reinterpret_cast<u32x4 *>(lds + c * 16) = *reinterpret_cast<const u32x4 *>(P.pack + (size_t)c * 16);
    } else if constexpr (LAYER == 1) {
#pragma unroll
        for (int s = 0; s < KS; s++) { Bh1[s] = glb16(bp + (size_t)s * 2048); Bl1[s] = glb16(bp + (size_t)s * 2048 + 16); }
        // opaque to the compiler from here on: otherwise it re-loads fragments inside the band loop instead of keeping them,
        // and (loads return in order) every band waits for its window prefetch in front of the first MFMA
#pragma unroll
        for (int s = 0; s < KS; s++) { asm volatile("" : "+v"(Bh1[s])); asm volatile("" : "+v"(Bl1[s])); }
    }
    const int ch = 32 * nh + m;
    float bias = P.bias ? P.bias[ch] : 0.f;
    float inv_scale = I8 ? P.inv1c[ch] : P.inv_scale;
    asm volatile("" : "+v"(bias), "+v"(inv_scale));        // waited for here, not at its first use behind the window prefetch
    bool ovf = false;
    const int band_end = min(P.nbands, ((int)blockIdx.y + 1) * P.bands_per_wg);
    // Staging is split in two: `issue` sends every load of a band's input window (all of a lane's loads before anything else,
    // so the window costs one memory round trip), `commit` converts / stores them to LDS.  Layer 1 issues the NEXT band's
    // loads before this band's MFMAs and commits them after the pooled rows have been stored; layers 2-3 issue them behind
    // the k-loop (loads return in order: in front of it a window prefetch would stall the weight ring).
    constexpr int ITER = LAYER == 1 ? (NR * NC + 127) / 128 : (NR * NC * 16 + 127) / 128;
    static_assert(LAYER == 1 ? 2 * ITER <= 32 : ITER <= 64, "one flag word per lane");
    uint32_t raw1[LAYER == 1 ? ITER : 1][2];
    u32x4 raw[LAYER == 1 ? 1 : ITER];
    uint32_t flags = 0u;                                      // layer 1, per staged pixel: bit 0 inside the frame, bit 1 the batch's last pixel
    uint64_t inside = 0u;                                     // layers 2-3, per staged 16-byte cell: inside the map
    auto issue = [&](int band) {
        const int iy0 = S * (2 * band * R - PPAD) - PAD;
        if (LAYER == 1) {
            flags = 0u;
        } else {
            inside = 0u;
        }
        if (LAYER == 1) {
#pragma unroll
            for (int it = 0; it < ITER; it++) {
                const int pix = tid + 128 * it, yy = pix / NC, y = iy0 + yy, x = ix0 + pix - yy * NC;
                const bool in = pix < NR * NC && y >= 0 && y < P.Hin && x >= 0 && x < P.Win;
                const uint8_t *src = P.img + (((size_t)img * P.Hin + (in ? y : 0)) * P.Win + (in ? x : 0)) * P.Cimg;
                // one unaligned dword (two for 6 channels) per pixel, branch-free and untouched until commit() so that all of a
                // lane's loads are in flight together; the batch's very last pixel reads the dwords that END at its last byte
                const bool tail = img == (int)gridDim.z - 1 && y == P.Hin - 1 && x == P.Win - 1;
                __builtin_memcpy(&raw1[it][0], src - (tail ? (CPIX == 4 ? 1 : 2) : 0), 4);
                if (CPIX == 8) __builtin_memcpy(&raw1[it][1], src + (tail ? 2 : 4), 4);
                flags |= (in ? 1u : 0u) << (2 * it) | (tail ? 2u : 0u) << (2 * it);
            }
        } else {
#pragma unroll
            for (int it = 0; it < ITER; it++) {
                const int idx = tid + 128 * it, pix = idx >> 4, pl = (idx >> 3) & 1, c8 = idx & 7;
                const int yy = pix / NC, y = iy0 + yy, x = ix0 + pix - yy * NC;
                // branch-free: cells outside the map read the frame's first pixel and are zeroed in commit()
                const bool in = pix < NR * NC && y >= 0 && y < P.Hin && x >= 0 && x < P.Win;
#if EG_X == 3
                raw[it] = u32x4{(uint32_t)idx, 0u, 0u, 0u};
#else
                raw[it] = *reinterpret_cast<const u32x4 *>((pl ? P.in_lo : P.in_hi) + (((size_t)img * P.Hin + (in ? y : 0)) * P.Win + (in ? x : 0)) * 64 + c8 * 8);
#endif
                inside |= (uint64_t)(in ? 1u : 0u) << it;
            }
        }
    };
    auto commit = [&]() {
        if (LAYER == 1) {
#pragma unroll
            for (int it = 0; it < ITER; it++) {
                const int pix = tid + 128 * it;
                if (pix < NR * NC) {
                    const bool in = (flags >> (2 * it)) & 1u, tail = (flags >> (2 * it)) & 2u;
                    uint32_t a, b = 0u;
                    if (CPIX == 4) a = tail ? raw1[it][0] >> 8 : raw1[it][0];
                    else {
                        const uint32_t w0 = raw1[it][0], w1 = raw1[it][1];
                        const uint32_t lo = tail ? (w0 >> 16) | (w1 << 16) : w0, hi = tail ? w1 >> 16 : w1;      // channels 0..3 | 4, 5
                        a = lo; b = (lo >> 24) | ((hi & 0xffffu) << 8);
                    }
                    if (I8) {
                        // (R - 128, G - 128, B - 128, mask 127) as four i8: one XOR, no conversion; outside the frame all zero
                        *reinterpret_cast<uint32_t *>(win + pix * 4) = in ? ((a & 0xffffffu) ^ 0x808080u) | ((uint32_t)kMaskI8 << 24) : 0u;
                        continue;
                    }
#if EG_X == 6
                    if (CPIX == 4) { *reinterpret_cast<uint32_t *>(win + pix * 8) = a; continue; }
#endif
                    a = in ? (a & 0xffffffu) | 0x01000000u : 0u;                           // byte 3: the mask
                    b = in ? b : 0u;
                    if (CPIX == 4) {
                        const half4v v = {(_Float16)(float)(a & 255u), (_Float16)(float)((a >> 8) & 255u), (_Float16)(float)((a >> 16) & 255u), (_Float16)(float)(a >> 24)};
                        *reinterpret_cast<half4v *>(win + pix * 8) = v;
                    } else {
                        const half8 v = {(_Float16)(float)(a & 255u), (_Float16)(float)((a >> 8) & 255u), (_Float16)(float)((a >> 16) & 255u),
                                         (_Float16)(float)(b & 255u), (_Float16)(float)((b >> 8) & 255u), (_Float16)(float)((b >> 16) & 255u),
                                         (_Float16)(float)(a >> 24), (_Float16)0.f};
                        *reinterpret_cast<half8 *>(win + pix * 16) = v;
                    }
                }
            }
        } else {
#pragma unroll
            for (int it = 0; it < ITER; it++) {
                const int idx = tid + 128 * it, pix = idx >> 4, pl = (idx >> 3) & 1, c8 = idx & 7;
                if (pix < NR * NC) *reinterpret_cast<u32x4 *>(win + pl * PLANE + pix * PX + c8 * 16) = ((inside >> it) & 1u) ? raw[it] : u32x4{0u, 0u, 0u, 0u};
            }
        }
    };
    const int band0 = blockIdx.y * P.bands_per_wg;
    // (the wide layer-3 window is 41 loads per lane: held across the pooling they spill; that variant keeps its loads at the loop top)
    constexpr bool PREFETCH = LAYER == 1 || LAYER == 2 || NC == kNC3Narrow;
    if (PREFETCH && band0 < band_end) issue(band0);
    for (int band = band0; band < band_end; band++) {
        const int p0 = band * R, rr0 = 2 * p0 - PPAD;
        __syncthreads();                                      // the previous band's fragments have been read
        if (!PREFETCH) issue(band);
        commit();
        __syncthreads();
        // ---- 2R+1 convolution rows x 32 columns x this wavefront's 32 channels: raw accumulators ----------------------------
        f32x16 acc[T];
#pragma unroll
        for (int t = 0; t < T; t++)
#pragma unroll
            for (int i = 0; i < 16; i++) acc[t][i] = 0.f;
        const int abase = woff + (LAYER == 1 ? (I8 ? (2 * m + 4 * h) * 4 : CPIX == 4 ? (2 * m + 2 * h) * 8 : (2 * m + h) * 16) : S * m * PX + h * 16);
        constexpr int ROWSTRIDE = S * NC * PIXB;
        if constexpr (I8) {
            if (band + 1 < band_end) issue(band + 1);
            // lane (m, h) of conv row t, kernel row ks: the 16 bytes of window pixels 2 m + 4 h .. + 3 (8-byte aligned: two 8-byte reads)
            auto frag = [&](int off) {
                const uint2 lo = *reinterpret_cast<const uint2 *>(lds + off), hi = *reinterpret_cast<const uint2 *>(lds + off + 8);
                i32x4 r; r[0] = (int)lo.x; r[1] = (int)lo.y; r[2] = (int)hi.x; r[3] = (int)hi.y;
                return r;
            };
            // Rows in pairs, the three digit chains of a pair together (six accumulators: consecutive MFMAs never share one), one pass over
            // the kernel rows per pair: a fragment is read from LDS once and meets its three digits.  All T rows at once would need 15
            // accumulators (240 registers); a digit at a time keeps every fragment of the pair live across the three passes (spills).
            i32x16 S[T];
#pragma unroll
            for (int t0 = 0; t0 < T; t0 += 2) {
                constexpr int G = 2;
                i32x16 ai[kI8Digits][G];
                // (opaque per pair: otherwise the digit loads of the three pairs are merged and all 21 fragments — 84 registers — stay live)
                int lb = lane;
                asm volatile("" : "+v"(lb));
#pragma unroll
                for (int d = 0; d < kI8Digits; d++)
#pragma unroll
                    for (int u = 0; u < G; u++)
#pragma unroll
                        for (int i = 0; i < 16; i++) ai[d][u][i] = 0;
#pragma unroll
                for (int ks = 0; ks < kI8Steps; ks++) {
                    i32x4 a[G];
#pragma unroll
                    for (int u = 0; u < G; u++) if (t0 + u < T) a[u] = frag(abase + (t0 + u) * ROWSTRIDE + ks * NC * 4);
#pragma unroll
                    for (int d = 0; d < kI8Digits; d++) {
                        const i32x4 B = *reinterpret_cast<const i32x4 *>(lds + ((size_t)((nh * kI8Steps + ks) * kI8Digits + d) * 64 + lb) * 16);
#pragma unroll
                        for (int u = 0; u < G; u++) if (t0 + u < T) ai[d][u] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[u], B, ai[d][u], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int u = 0; u < G; u++)
                    if (t0 + u < T) {
#pragma unroll
                        for (int i = 0; i < 16; i++) S[t0 + u][i] = ai[0][u][i] * 256 + ai[1][u][i] + (ai[2][u][i] >> 8);
                    }
                __builtin_amdgcn_sched_barrier(0);            // the scheduler may not merge the pairs' passes (all their accumulators live at once)
            }
            // (int -> float is monotone: the float max-pool below picks the same winners; |S| < 2^30 rounds to 24 bits like any f32 sum)
#pragma unroll
            for (int t = 0; t < T; t++)
#pragma unroll
                for (int i = 0; i < 16; i++) acc[t][i] = (float)S[t][i];
        } else if (LAYER == 1) {
            if (band + 1 < band_end) issue(band + 1);
#pragma unroll
            for (int s = 0; s < KS; s++) {
                const int koff = CPIX == 4 ? ((s >> 1) * NC + 4 * (s & 1)) * 8 : ((s >> 2) * NC + 2 * (s & 3)) * 16;
                half8 a[T];                                   // consecutive MFMAs go to different accumulators
#pragma unroll
                for (int t = 0; t < T; t++) a[t] = lds16(abase + t * ROWSTRIDE + koff);
                const half8 Bh = LDSB ? lds16(((nh * KS + s) * 64 + lane) * 32) : Bh1[LDSB ? 0 : s];
                const half8 Bl = LDSB ? lds16(((nh * KS + s) * 64 + lane) * 32 + 16) : Bl1[LDSB ? 0 : s];
#if EG_X == 4
#pragma unroll
                for (int t = 0; t < T; t++) { asm volatile("" :: "v"(a[t]), "v"(Bh)); asm volatile("" :: "v"(Bl)); }
#else
#pragma unroll
                for (int t = 0; t < T; t++) acc[t] = mfma16(a[t], Bh, acc[t]);
#pragma unroll
                for (int t = 0; t < T; t++) acc[t] = mfma16(a[t], Bl, acc[t]);
#endif
            }
        } else {
            // Round 6: the k-loop as a hand-scheduled stream, like the fused kernel's layer 2 (csrc/encoder.hip, NOTES section O).  Left to
            // the compiler it read a fragment, waited out the LDS round trip and issued its MFMAs — two of them back to back on the
            // same accumulator.  Every MFMA is a pinned statement now: per k-step 3 T of them — hi x hi, hi x lo, lo x hi for each of
            // the T rows, consecutive ones on different accumulators, the same order per accumulator as before (bit-identical sums) —
            // with the NEXT k-step's 2 T fragments requested one per gap and the weight ring's two refills behind their last use.
            // 36 k-steps fully unrolled: ring slots, fragment rotation and tap offsets are compile-time.
            half8 rbh[8], rbl[8];
#pragma unroll
            for (int u = 0; u < 8; u++) { rbh[u] = glb16(bp + (size_t)u * 2048); rbl[u] = glb16(bp + (size_t)u * 2048 + 16); }
            auto koff = [&](int ks) { const int tap = ks >> 2, ky = tap / 3, kx = tap - 3 * ky; return abase + (ky * NC + kx) * PX + (ks & 3) * 32; };
            // ONE running pointer for the ring's refills, opaque per k-step: 28 distinct 64-bit addresses would otherwise be computed ahead
            // and parked in AGPRs (56 registers the window prefetch needs)
            const char *rp = bp + (size_t)8 * 2048;
            half8 ah[T], al[T], nh_[T], nl_[T];
#pragma unroll
            for (int t = 0; t < T; t++) { ah[t] = lds16(koff(0) + t * ROWSTRIDE); al[t] = lds16(PLANE + koff(0) + t * ROWSTRIDE); nh_[t] = ah[t]; nl_[t] = al[t]; }
#pragma unroll
            for (int ks = 0; ks < 36; ks++) {
                const int slot = ks & 7;
                const half8 Bh = rbh[slot], Bl = rbl[slot];
#pragma unroll
                for (int t = 0; t < T; t++) {
#if EG_X != 1
                    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[t]) : "v"(ah[t]), "v"(Bh));
#else
                    asm volatile("" :: "v"(ah[t]), "v"(Bh));
#endif
                    if (ks + 1 < 36) nh_[t] = lds16(koff(ks + 1) + t * ROWSTRIDE);
                }
#pragma unroll
                for (int t = 0; t < T; t++) {
#if EG_X != 1
                    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[t]) : "v"(ah[t]), "v"(Bl));
#else
                    asm volatile("" :: "v"(ah[t]), "v"(Bl));
#endif
                    if (ks + 1 < 36) nl_[t] = lds16(PLANE + koff(ks + 1) + t * ROWSTRIDE);
                }
#pragma unroll
                for (int t = 0; t < T; t++) {
#if EG_X != 1
                    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[t]) : "v"(al[t]), "v"(Bh));
#else
                    asm volatile("" :: "v"(al[t]), "v"(Bh));
#endif
                    if (t == 0 && ks + 8 < 36) rbh[slot] = glb16(rp);
                    if (t == 1 && ks + 8 < 36) rbl[slot] = glb16(rp + 16);
                }
                if (ks + 8 < 36) { rp += 2048; asm volatile("" : "+v"(rp)); }
#pragma unroll
                for (int t = 0; t < T; t++) { ah[t] = nh_[t]; al[t] = nl_[t]; }
            }
            // XDL write -> VALU read distance of the pinned MFMAs (the hazard recogniser does not see inside the statements)
            if constexpr (T == 5) asm volatile("s_nop 15\n\ts_nop 3" : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]), "+a"(acc[3]), "+a"(acc[4]));
            else if constexpr (T == 3) asm volatile("s_nop 15\n\ts_nop 3" : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]));
            else static_assert(T == 3 || T == 5, "fence the accumulators of this band height");
        }
        // layers 2-3 (round 6): the NEXT band's window is requested HERE — behind the last weight-ring load of this band (loads return in
        // order: requested earlier it would stall the ring) and in front of the pooling, the band's stores and the loop turn-around, which
        // now cover its memory round trip instead of following it (one wavefront per SIMD: nothing else would)
        if (LAYER != 1 && PREFETCH && band + 1 < band_end) issue(band + 1);
        // ---- 3x3/2 max-pool on the raw accumulators, then scale / bias / ReLU on the pooled values only --------------------------
        // acc[t][i] of lane l: convolution row rr0 + t, column c0 + 8 (i / 4) + 4 (l / 32) + i % 4, channel ch.
        // The pooled band is assembled in LDS (over the input window, which is dead by now) as [plane][row][column][64 channels]
        // and leaves with 16-byte stores: one 128-byte line per pixel and plane.
        __syncthreads();
        constexpr int OPIX = R * 15, OPLANE = OPIX * 128;
        // pooled value `val` of row j, column jj of the band -> scale / bias / ReLU -> the LDS band (hi | lo planes, or float32)
        auto emit = [&](int j, int jj, float val, bool on) {
            const float x = fmaxf(val * inv_scale + bias, 0.f);
            if (on) {
                if (LAYER == 3) *reinterpret_cast<float *>(win + ((j * 15 + jj) * 64 + ch) * 4) = x;
                else {
                    ovf |= !(x < kF16Max);                    // columns beyond the map hold -inf -> 0 after the ReLU: never flagged
                    const _Float16 hi = (_Float16)x;
                    *reinterpret_cast<_Float16 *>(win + ((j * 15 + jj) * 64 + ch) * 2) = hi;
                    *reinterpret_cast<_Float16 *>(win + OPLANE + ((j * 15 + jj) * 64 + ch) * 2) = (_Float16)(x - (float)hi);
                }
            }
        };
#if EG_X == 5
        if (LAYER == 1) { float sink = 0.f; for (int t = 0; t < T; t++) sink += acc[t][0]; if (sink == 12345.f) ovf = true; __syncthreads(); continue; }
#endif
#if EG_X == 2
        if (LAYER != 1) { float sink = 0.f; for (int t = 0; t < T; t++) sink += acc[t][0]; if (sink == 12345.f) ovf = true; __syncthreads(); continue; }
#endif
        const bool edge = c0 < 0 || c0 + 32 > P.Wc;           // the map's left / right edge runs through this tile
        // Rows in pairs: lanes 0-31 finish pooled row 2jp, lanes 32-63 row 2jp + 1.  Columns arrive eight at a time
        // (accumulator quad g of both wavefront halves): v_permlane32_swap(va, vb) hands every lane its own row's value of the
        // low quad (first result) and of the high quad (second result); a pooled column is emitted as soon as its three
        // inputs exist, so only nine column values are live at once and no lane idles in a divergent branch.
#pragma unroll
        for (int jp = 0; jp < R / 2; jp++) {
            const int ta = 4 * jp, jmy = 2 * jp + h;
            if (p0 + 2 * jp >= P.Hp) break;
            bool rv[5];
#pragma unroll
            for (int k = 0; k < 5; k++) rv[k] = rr0 + ta + k >= 0 && rr0 + ta + k < P.Hc;
            const bool on = p0 + jmy < P.Hp;
            float prev6 = kNegInf, prev7 = kNegInf;
#pragma unroll
            for (int g = 0; g < 4; g++) {
                float c[8];
#pragma unroll
                for (int i4 = 0; i4 < 4; i4++) {
                    const int i = 4 * g + i4;
                    const float mid = rv[2] ? acc[ta + 2][i] : kNegInf;
                    const float va = max3(rv[0] ? acc[ta][i] : kNegInf, rv[1] ? acc[ta + 1][i] : kNegInf, mid);
                    const float vb = max3(mid, rv[3] ? acc[ta + 3][i] : kNegInf, rv[4] ? acc[ta + 4][i] : kNegInf);
                    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(va), __float_as_uint(vb), false, false);
                    c[i4] = __uint_as_float(sw[0]);
                    c[4 + i4] = __uint_as_float(sw[1]);
                }
                if (edge) {
#pragma unroll
                    for (int k = 0; k < 8; k++)
                        if (c0 + 8 * g + k < 0 || c0 + 8 * g + k >= P.Wc) c[k] = kNegInf;
                }
                if (g > 0) emit(jmy, 4 * g - 1, max3(prev6, prev7, c[0]), on);
                emit(jmy, 4 * g, max3(c[0], c[1], c[2]), on);
                emit(jmy, 4 * g + 1, max3(c[2], c[3], c[4]), on);
                emit(jmy, 4 * g + 2, max3(c[4], c[5], c[6]), on);
                prev6 = c[6]; prev7 = c[7];
            }
        }
        // an odd last row: both halves hold all 32 columns (v_permlane32_swap(v, v)) and share the pooled columns between them
        if (R & 1) {
            constexpr int j = R - 1;
            if (p0 + j < P.Hp) {
                const bool rv0 = rr0 + 2 * j >= 0 && rr0 + 2 * j < P.Hc, rv1 = rr0 + 2 * j + 1 < P.Hc, rv2 = rr0 + 2 * j + 2 < P.Hc;
                float prev6 = kNegInf, prev7 = kNegInf;
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    float c[8];
#pragma unroll
                    for (int i4 = 0; i4 < 4; i4++) {
                        const int i = 4 * g + i4;
                        const float v = max3(rv0 ? acc[2 * j][i] : kNegInf, rv1 ? acc[2 * j + 1][i] : kNegInf, rv2 ? acc[2 * j + 2][i] : kNegInf);
                        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
                        c[i4] = __uint_as_float(sw[0]);
                        c[4 + i4] = __uint_as_float(sw[1]);
                    }
                    if (edge) {
#pragma unroll
                        for (int k = 0; k < 8; k++)
                            if (c0 + 8 * g + k < 0 || c0 + 8 * g + k >= P.Wc) c[k] = kNegInf;
                    }
                    if (g > 0) emit(j, 4 * g - 1, max3(prev6, prev7, c[0]), ((4 * g - 1) & 1) == h);
                    emit(j, 4 * g, max3(c[0], c[1], c[2]), ((4 * g) & 1) == h);
                    emit(j, 4 * g + 1, max3(c[2], c[3], c[4]), ((4 * g + 1) & 1) == h);
                    emit(j, 4 * g + 2, max3(c[4], c[5], c[6]), ((4 * g + 2) & 1) == h);
                    prev6 = c[6]; prev7 = c[7];
                }
            }
        }
        __syncthreads();
        constexpr int CPP = LAYER == 3 ? 16 : 8;              // 16-byte chunks per pixel (and plane)
        constexpr int NCHUNK = OPIX * CPP * (LAYER == 3 ? 1 : 2);
#pragma unroll
        for (int it = 0; it < (NCHUNK + 127) / 128; it++) {
            const int c = tid + 128 * it, pl = c / (OPIX * CPP), rem = c - pl * (OPIX * CPP), pix = rem / CPP, c8 = rem - pix * CPP;
            const int j = pix / 15, jj = pix - 15 * j, p = p0 + j, q = 15 * seg + jj;
            if (c < NCHUNK && p < P.Hp && q < P.Wp) {
                const u32x4 v = *reinterpret_cast<const u32x4 *>(win + c * 16);
                const size_t o = (((size_t)img * P.Hp + p) * P.Wp + q) * 64;
                if (LAYER == 3) *reinterpret_cast<u32x4 *>(P.out_f32 + o + c8 * 4) = v;
                else *reinterpret_cast<u32x4 *>((pl ? P.out_lo : P.out_hi) + o + c8 * 8) = v;
            }
        }
    }
    if (ovf) atomicOr(P.status, 1);
}

// out[frame][s] = fcb[s] + sum_f feat[frame][f] * w[s][f]; one workgroup per frame, one output per wavefront at a time
__global__ __launch_bounds__(256) void enc_fc_k(const float *feat, const float *w, const float *b, float *out, int F, int state_dim) {
    const int img = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float *x = feat + (size_t)img * F;
    for (int s = wave; s < state_dim; s += 4) {
        const float *ws = w + (size_t)s * F;
        double a = 0.0;
        for (int f = lane; f < F; f += 64) a += (double)x[f] * (double)ws[f];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) a += __shfl_xor(a, d);
        if (lane == 0) out[(size_t)img * state_dim + s] = (float)(a + (double)b[s]);
    }
}

// layer 1, normalisation folded: staged pixel = (c_0 .. c_{C-1}, mask [, 0]); k = (ky * 8 + kx) * CPIX + c with kx = 7 a zero slot.
// The mask channel carries -sum_c w * mean_c / std_c per tap (zero padding stays exact in NORMALISED space) and the folded bias
// on the centre tap.  The frame's axes are swapped with respect to torch's, so the tap at frame offset (ky, kx) is w[o][c][kx][ky].
double layer1_weight(const float *w, const float *b, int C, int cpix, int o, int k) {
    const int ky = k / (8 * cpix), kx = (k / cpix) % 8, c = k % cpix;
    if (kx >= 7) return 0.0;
    if (c < C) return (double)w[((o * C + c) * 7 + kx) * 7 + ky] / (255.0 * (double)kStd[c % 3]);
    if (c > C) return 0.0;
    double v = 0.0;
    for (int cc = 0; cc < C; cc++) v -= (double)w[((o * C + cc) * 7 + kx) * 7 + ky] * (double)kMean[cc % 3] / (double)kStd[cc % 3];
    if (ky == 3 && kx == 3) v += (double)b[o];
    return v;
}
float pack_layer1(const float *w, const float *b, int C, _Float16 *out) {
    const int cpix = C == 3 ? 4 : 8, ks = 7 * 8 * cpix / 16;
    double wmax = 0.0;
    for (int o = 0; o < 64; o++)
        for (int k = 0; k < 16 * ks; k++) wmax = fmax(wmax, fabs(layer1_weight(w, b, C, cpix, o, k)));
    const float scale = pick_scale(wmax);
    for (int nh = 0; nh < 2; nh++)
        for (int s = 0; s < ks; s++)
            for (int lane = 0; lane < 64; lane++)
                for (int e = 0; e < 8; e++) {
                    const int o = 32 * nh + (lane & 31), k = 16 * s + 8 * (lane >> 5) + e;
                    _Float16 *dst = out + ((size_t)(nh * ks + s) * 64 + lane) * 16;
                    split_f16((float)(layer1_weight(w, b, C, cpix, o, k) * (double)scale), dst[e], dst[8 + e]);
                }
    return scale;
}

}  // namespace

float pack_layer1_general(const float *w, const float *b, int n_channels, void *out, size_t out_bytes) {
    if (!w || !b || !out || (n_channels != 3 && n_channels != 6)) return 0.f;
    const size_t need = (size_t)2 * (n_channels == 3 ? 14 : 28) * 2048;
    if (out_bytes < need) return 0.f;
    return pack_layer1(w, b, n_channels, static_cast<_Float16 *>(out));
}

struct General {
    int device_id, state_dim, F;
    Geometry g;
    char *d_pack;          // layer 1 | layer 2 | layer 3 B fragments
    char *d_pack_i8;       // 3-channel frames: layer 1 as int8 digits (pack_layer1_i8, zero slot 7), then [64] float 256 / scale
    bool l1_i8;            // layer 1 runs on the int8 pipe (3 channels, SRLHIP_ENCODER_L1=i8: the measured alternative, slower here)
    size_t pack1_bytes;
    float *d_f32;          // bias2[64] bias3[64] fcb[state_dim] fcw[state_dim][F]
    float inv_scale[3];
    int *d_status;
    char *d_act;           // scratch for `cap` frames: a1 hi | a1 lo | a2 hi | a2 lo | features
    int cap;
};

int *general_status(General *g) { return g->d_status; }

void general_destroy(General *g) {
    if (!g) return;
    (void)hipSetDevice(g->device_id);
    if (g->d_pack) (void)hipFree(g->d_pack);
    if (g->d_pack_i8) (void)hipFree(g->d_pack_i8);
    if (g->d_f32) (void)hipFree(g->d_f32);
    if (g->d_status) (void)hipFree(g->d_status);
    if (g->d_act) (void)hipFree(g->d_act);
    delete g;
}

int general_create(int device_id, const Geometry &geo, int state_dim, const float *conv1_w, const float *conv1_b, const float *conv2_w,
                   const float *conv2_b, const float *conv3_w, const float *conv3_b, const float *fc_w, const float *fc_b,
                   General **out, std::string &err) {
    General *g = new (std::nothrow) General();
    if (!g) return SRLHIP_ENOMEM;
    g->device_id = device_id; g->state_dim = state_dim; g->g = geo; g->d_pack = nullptr; g->d_f32 = nullptr; g->d_status = nullptr;
    g->d_act = nullptr; g->cap = 0; g->d_pack_i8 = nullptr;
    {
        // The int8 form of layer 1 is NOT the default here (it is in the fused 64x64x3 kernel): left to the compiler's scheduler its six
        // digit accumulators per row pair + the 80 recombined sums + hoisted fragment loads spill (88 registers at the 256 of two
        // workgroups per CU, 50 at 512), and 4096 frames of 224x224 take 13.4 ms against 8.2 ms (profiles/NOTES.md section U).
        // SRLHIP_ENCODER_L1=i8 selects it (tests run both).
        const char *l1 = getenv("SRLHIP_ENCODER_L1");
        g->l1_i8 = geo.C == 3 && l1 && strcmp(l1, "i8") == 0;
    }
    const int F = 64 * geo.Hp[2] * geo.Wp[2];
    g->F = F;
    const int ks1 = geo.C == 3 ? 14 : 28;
    g->pack1_bytes = (size_t)2 * ks1 * 64 * 32;
    std::vector<char> pack(g->pack1_bytes + 2 * kPack3x3Bytes);
    const float s1 = pack_layer1(conv1_w, conv1_b, geo.C, reinterpret_cast<_Float16 *>(pack.data()));
    const float s2 = pack_layer3x3(conv2_w, reinterpret_cast<_Float16 *>(pack.data() + g->pack1_bytes));
    const float s3 = pack_layer3x3(conv3_w, reinterpret_cast<_Float16 *>(pack.data() + g->pack1_bytes + kPack3x3Bytes));
    g->inv_scale[0] = 1.f / s1; g->inv_scale[1] = 1.f / s2; g->inv_scale[2] = 1.f / s3;      // powers of two: exact
    std::vector<float> f(128 + (size_t)state_dim * (1 + F));
    memcpy(f.data(), conv2_b, 64 * sizeof(float));
    memcpy(f.data() + 64, conv3_b, 64 * sizeof(float));
    memcpy(f.data() + 128, fc_b, state_dim * sizeof(float));
    // features are [frame row p][frame column q][channel]; torch flattens the transposed map as (channel, q, p)
    const int Hp = geo.Hp[2], Wp = geo.Wp[2];
    for (int s = 0; s < state_dim; s++)
        for (int p = 0; p < Hp; p++)
            for (int q = 0; q < Wp; q++)
                for (int c = 0; c < 64; c++)
                    f[128 + state_dim + (size_t)s * F + (p * Wp + q) * 64 + c] = fc_w[(size_t)s * F + (size_t)c * Wp * Hp + q * Hp + p];
#define GEN_CHECK(expr)                                                              \
    do {                                                                             \
        hipError_t e__ = (expr);                                                     \
        if (e__ != hipSuccess) {                                                     \
            err = std::string(#expr ": ") + hipGetErrorString(e__);                  \
            general_destroy(g);                                                      \
            return SRLHIP_EHIP;                                                      \
        }                                                                            \
    } while (0)
    GEN_CHECK(hipSetDevice(device_id));
    GEN_CHECK(hipMalloc(reinterpret_cast<void **>(&g->d_pack), pack.size()));
    GEN_CHECK(hipMalloc(reinterpret_cast<void **>(&g->d_f32), f.size() * sizeof(float)));
    GEN_CHECK(hipMalloc(reinterpret_cast<void **>(&g->d_status), sizeof(int)));
    GEN_CHECK(hipMemcpy(g->d_pack, pack.data(), pack.size(), hipMemcpyHostToDevice));
    GEN_CHECK(hipMemcpy(g->d_f32, f.data(), f.size() * sizeof(float), hipMemcpyHostToDevice));
    GEN_CHECK(hipMemset(g->d_status, 0, sizeof(int)));
    if (geo.C == 3) {
        std::vector<char> p8(kPackI8Bytes + 64 * sizeof(float));
        pack_layer1_i8(conv1_w, conv1_b, 7, reinterpret_cast<int8_t *>(p8.data()), reinterpret_cast<float *>(p8.data() + kPackI8Bytes));
        GEN_CHECK(hipMalloc(reinterpret_cast<void **>(&g->d_pack_i8), p8.size()));
        GEN_CHECK(hipMemcpy(g->d_pack_i8, p8.data(), p8.size(), hipMemcpyHostToDevice));
        GEN_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(enc_layer_k<1, 4, kR1, kNC1, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    }
    constexpr int lds2 = 2 * (kR2 * 2 + 3) * kNC2 * PX, lds3 = 2 * (2 * (2 * kR3) + 3) * kNC3 * PX;
    GEN_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(enc_layer_k<2, 4, kR2, kNC2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds2));
    GEN_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(enc_layer_k<1, 4, kR1, kNC1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    GEN_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(enc_layer_k<3, 4, kR3, kNC3>), hipFuncAttributeMaxDynamicSharedMemorySize, lds3));
    GEN_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(enc_layer_k<3, 4, kR3, kNC3Narrow>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  2 * (2 * (2 * kR3) + 3) * kNC3Narrow * PX + kLdsSlack3));
#undef GEN_CHECK
    *out = g;
    return SRLHIP_OK;
}

// dynamic LDS of a layer kernel: the input window, or the pooled band that replaces it before the stores
static int lds_bytes(int window, int r, bool f32) { const int band = r * 15 * (f32 ? 256 : 256); return window > band ? window : band; }

// layers 2-3: bands a workgroup walks — as many as leave >= 4096 workgroups
static int bands_per_wg(int nbands, int nsegs, int frames) {
    const long long wgs = (long long)nbands * nsegs * frames;
    const long long b = wgs / 4096;
    return (int)(b < 1 ? 1 : (b > nbands ? nbands : b));
}

int general_forward(General *g, const uint8_t *images_dev, int n, float *states_dev, hipStream_t stream, std::string &err) {
    const Geometry &geo = g->g;
    const size_t a1 = (size_t)geo.Hp[0] * geo.Wp[0] * 64, a2 = (size_t)geo.Hp[1] * geo.Wp[1] * 64;      // halfs per frame and plane
    const size_t per_frame = 2 * a1 * 2 + 2 * a2 * 2 + (size_t)g->F * 4;
    if (n > g->cap) {
        // grow-only scratch; a caller that captures the forward into a hipGraph warms the handle up with its batch size first
        if (g->d_act) { (void)hipStreamSynchronize(stream); (void)hipFree(g->d_act); g->d_act = nullptr; g->cap = 0; }
        hipError_t rc = hipMalloc(reinterpret_cast<void **>(&g->d_act), per_frame * (size_t)n + 256);
        if (rc != hipSuccess) { err = std::string("encoder scratch hipMalloc: ") + hipGetErrorString(rc); return SRLHIP_ENOMEM; }
        g->cap = n;
    }
    const size_t cap = (size_t)g->cap;
    _Float16 *a1h = reinterpret_cast<_Float16 *>(g->d_act), *a1l = a1h + a1 * cap, *a2h = a1l + a1 * cap, *a2l = a2h + a2 * cap;
    float *feat = reinterpret_cast<float *>(a2l + a2 * cap);
    const char *pack2 = g->d_pack + g->pack1_bytes, *pack3 = pack2 + kPack3x3Bytes;
    constexpr int kChunk = 32768;                            // frames per launch (grid.z)
    for (int base = 0; base < n; base += kChunk) {
        const int nn = n - base < kChunk ? n - base : kChunk;
        LayerParams p = {};
        p.status = g->d_status;
        // layer 1
        p.img = images_dev + (size_t)base * geo.H * geo.W * geo.C;
        p.out_hi = a1h + a1 * base; p.out_lo = a1l + a1 * base; p.pack = g->d_pack; p.bias = nullptr; p.inv_scale = g->inv_scale[0];
        p.Hin = geo.H; p.Win = geo.W; p.Cimg = geo.C; p.Hc = geo.Hc[0]; p.Wc = geo.Wc[0]; p.Hp = geo.Hp[0]; p.Wp = geo.Wp[0];
        const int r1 = geo.C == 3 ? kR1 : kR1x6;
        // layer 1 walks 4..8 bands per workgroup with its weights resident: the count that leaves the last workgroup least idle
        p.nbands = (p.Hp + r1 - 1) / r1; p.bands_per_wg = 4;
        for (int b = 5; b <= 8; b++)
            if ((p.nbands + b - 1) / b * b - p.nbands <= (p.nbands + p.bands_per_wg - 1) / p.bands_per_wg * p.bands_per_wg - p.nbands) p.bands_per_wg = b;
        dim3 grid1((p.Wp + 14) / 15, (p.nbands + p.bands_per_wg - 1) / p.bands_per_wg, nn);
        if (geo.C == 3 && g->l1_i8) {
            // 3 channels, int8 layer 1: digits in LDS, four wavefronts per workgroup (two column segments), bands of kR1 pooled rows
            constexpr int win = (2 * (2 * kR1) + 7) * kNC1 * 4, band = kR1 * 15 * 256;
            p.pack = g->d_pack_i8; p.inv1c = reinterpret_cast<const float *>(g->d_pack_i8 + kPackI8Bytes);
            hipLaunchKernelGGL((enc_layer_k<1, 4, kR1, kNC1, true, true>), dim3((grid1.x + 1) / 2, grid1.y, nn), dim3(256),
                               (int)kPackI8Bytes + 2 * (win > band ? win : band), stream, p);
        } else if (geo.C == 3) {
            // 3 channels: weights in LDS, four wavefronts per workgroup (two column segments), bands of kR1 pooled rows
            constexpr int win = (2 * (2 * kR1) + 7) * kNC1 * 8, band = kR1 * 15 * 256;
            hipLaunchKernelGGL((enc_layer_k<1, 4, kR1, kNC1, true>), dim3((grid1.x + 1) / 2, grid1.y, nn), dim3(256),
                               2 * 14 * 2048 + 2 * (win > band ? win : band), stream, p);
        } else {
            hipLaunchKernelGGL((enc_layer_k<1, 8, kR1x6, kNC1>), grid1, dim3(128), lds_bytes((2 * (2 * kR1x6) + 7) * kNC1 * 16, kR1x6, false), stream, p);
        }
        // layer 2
        p.img = nullptr; p.in_hi = a1h + a1 * base; p.in_lo = a1l + a1 * base; p.out_hi = a2h + a2 * base; p.out_lo = a2l + a2 * base;
        p.pack = pack2; p.bias = g->d_f32; p.inv_scale = g->inv_scale[1];
        p.Hin = geo.Hp[0]; p.Win = geo.Wp[0]; p.Hc = geo.Hc[1]; p.Wc = geo.Wc[1]; p.Hp = geo.Hp[1]; p.Wp = geo.Wp[1];
        p.nbands = (p.Hp + kR2 - 1) / kR2; p.bands_per_wg = bands_per_wg(p.nbands, (p.Wp + 14) / 15, nn);
        hipLaunchKernelGGL((enc_layer_k<2, 4, kR2, kNC2>), dim3((p.Wp + 14) / 15, (p.nbands + p.bands_per_wg - 1) / p.bands_per_wg, nn), dim3(128), 2 * (kR2 * 2 + 3) * kNC2 * PX, stream, p);
        // layer 3
        p.in_hi = a2h + a2 * base; p.in_lo = a2l + a2 * base; p.out_hi = nullptr; p.out_lo = nullptr; p.out_f32 = feat + (size_t)g->F * base;
        p.pack = pack3; p.bias = g->d_f32 + 64; p.inv_scale = g->inv_scale[2];
        p.Hin = geo.Hp[1]; p.Win = geo.Wp[1]; p.Hc = geo.Hc[2]; p.Wc = geo.Wc[2]; p.Hp = geo.Hp[2]; p.Wp = geo.Wp[2];
        p.nbands = (p.Hp + kR3 - 1) / kR3; p.bands_per_wg = bands_per_wg(p.nbands, (p.Wp + 14) / 15, nn);
        if (p.Wc <= 16)
            hipLaunchKernelGGL((enc_layer_k<3, 4, kR3, kNC3Narrow>), dim3(1, (p.nbands + p.bands_per_wg - 1) / p.bands_per_wg, nn), dim3(128),
                               2 * (2 * (2 * kR3) + 3) * kNC3Narrow * PX + kLdsSlack3, stream, p);
        else
            hipLaunchKernelGGL((enc_layer_k<3, 4, kR3, kNC3>), dim3((p.Wp + 14) / 15, (p.nbands + p.bands_per_wg - 1) / p.bands_per_wg, nn), dim3(128),
                               2 * (2 * (2 * kR3) + 3) * kNC3 * PX, stream, p);
        hipLaunchKernelGGL(enc_fc_k, dim3(nn), dim3(256), 0, stream, feat + (size_t)g->F * base, g->d_f32 + 128 + g->state_dim, g->d_f32 + 128,
                           states_dev + (size_t)base * g->state_dim, g->F, g->state_dim);
    }
    const hipError_t rc = hipGetLastError();
    if (rc != hipSuccess) { err = std::string("encoder layer launch: ") + hipGetErrorString(rc); return SRLHIP_EHIP; }
    return SRLHIP_OK;
}

}  // namespace srlenc
