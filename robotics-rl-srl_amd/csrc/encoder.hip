// encoder.hip — fused SRL encoder forward (srl_zoo CustomCNN) for the raw_pixels path on gfx950.
//
// Replaces, for a whole device-resident batch of rasterised frames:
//   SRLNeuralNetwork.getState        state_representation/models.py:178-193  (preprocess + model forward)
//   MultiprocessSRLModel._run        rl_baselines/utils.py:181-191           (one image at a time behind queues)
// CustomCNN (restated from srl_zoo; BatchNorms folded by the host side):
//   conv7x7/2 p3 (3->64) + ReLU + maxpool3/2 p1 -> conv3x3 p1 (64->64) + ReLU + maxpool3/2
//   -> conv3x3/2 p1 (64->64) + ReLU + maxpool3/2 -> FC(64 -> state_dim)            at 64x64x3 input.
//
// MI355X design: ONE kernel, one 256-lane workgroup (one wave per SIMD) per CU looping over images; an image never leaves
// the CU between the uint8 frame (12 KiB read from HBM) and its state vector (state_dim floats written):
//   * every activation lives in LDS (148 KiB of the CU's 160 KiB); the three convolutions are implicit GEMMs on
//     v_mfma_f32_32x32x16_f16 with A = pixels (im2col fragments are contiguous 16-byte LDS reads: channels are
//     the fastest index, the 3-channel input is repacked to RGB+mask so a pixel is 8 bytes) and B = weights;
//   * f32 accuracy on the f16 matrix pipe (16x the f32 MFMA rate): every operand is split x = hi + lo with
//     hi = f16(x), lo = f16(x - hi); the products hi*hi, hi*lo and lo*hi accumulate in float32 inside the MFMA
//     (lo*lo ~ 2^-22 is dropped).  Each layer's weights are pre-scaled by a power of two chosen by the packer so
//     that the lo parts stay in f16's normal range; the scale is undone in the epilogue.  uint8 pixels are exact
//     in f16, so layer 1 needs only the weight split (2 MFMAs per fragment), layers 2-3 need 3;
//   * layer 1 (round 6) runs on the INT8 matrix pipe, exactly: the frame's bytes are the A operand as they are (p - 128 as i8: one XOR,
//     no f16 unpack; a pixel is 4 bytes R G B mask), every folded layer-1 weight is a per-output-channel 24-bit fixed-point number
//     split into three balanced base-256 digits, v_mfma_i32_32x32x32_i8 x 3 accumulate in int32 without rounding (K = 224 x 128 x 128
//     < 2^22), and the three sums are recombined in integer arithmetic in the epilogue — 21 MFMAs of K = 32 per conv row where the
//     f16 form takes 28 of K = 16, half the LDS fragment bytes, and the max-pool runs on int32.  The f16 layer 1 (rounds 2-5) is kept
//     as the measured alternative (SRLHIP_ENCODER_L1=f16) and for the two-waves-per-SIMD variant;
//   * the ImageNet normalisation of preprocessImage is folded into the layer-1 weights; the fourth input channel
//     is a validity mask (1 inside the image, 0 in the padding ring) whose weights carry -sum_c w*mean_c/std_c
//     per tap, which keeps zero padding in NORMALISED space exact; the folded BN bias rides on the centre tap;
//   * max-pool runs on the raw accumulator registers BEFORE scale / bias / ReLU (all monotone, the bias is per
//     channel): a lane owns one output channel, the 3x3 windows need only lane-local maxima plus 2-4 values from
//     lane^32, and only the pooled quarter of the map is rescaled, split and written back as f16 hi/lo planes with
//     a 144-byte pixel pitch (conflict-free ds_read_b128 for 16 consecutive pixels);
//   * B fragments come from L2 (352 KiB, shared by every CU): layer 1's (112 VGPRs per wave) are requested before
//     the frame is unpacked and stay in registers for the layer, layer 2's are streamed through a 6-deep register
//     ring, layer 3's arrive in one burst behind layer 2's k-loop; the next frame's bytes are fetched during layer 2.
//     Two waves per SIMD (MG = 4, SRLHIP_ENCODER_WAVES=8) is kept as a measured alternative: 0.55 ms (0.64 with its MFMAs padded, see mfma16_pinned) against 0.43 ms
//     per 4096 frames — each wave then streams its own B copies, 256 registers per lane spill the hoisted
//     addresses, and the groups wait for each other at the barriers.  Weights resident across frames starve the
//     layer-2 loop of registers (its ring becomes loop-carried copies behind `s_waitcnt vmcnt(0)`).
// The reference transposes H and W before the network (models.py:185-188); all layers are symmetric in the two
// spatial dims, so the kernel works on the frame as rasterised and the host packer swaps the two kernel axes.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <cmath>
#include <new>
#include <string>
#include <vector>

#include "../../include/srlhip.h"
#include "encoder_general.hpp"

#ifndef ENC_X
#define ENC_X 0          // experiment builds (profiles/probes/encoder_experiments.sh); 0 = the product
#endif

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4v __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

constexpr int kImg = 64, kCh = 64;
constexpr int kS1 = 14;                 // layer-1 k-steps: K = 7 rows x (8 pixel slots x 4 channels) = 224
constexpr int kS1i = srlenc::kI8Steps;  // int8 layer 1: one k-step per kernel row, K = 8 pixel slots x (R, G, B, mask) bytes = 32
constexpr int kDigits = srlenc::kI8Digits;   // balanced base-256 digits of a 24-bit fixed-point weight, most significant first
constexpr unsigned kMaskByte = (unsigned)srlenc::kMaskI8 << 24;   // the validity-mask byte of an inside pixel is 127 (its tap carries weight / 127: same scale as the colour taps)
constexpr int kIntNegInf = -2147483647 - 1;   // max-pool padding on combined int32 sums (|sum| <= 9.4e8)
constexpr int kS2 = 36;                 // layers 2/3:      K = 9 taps x 64 channels = 576
constexpr int kB2Ahead = 6;             // layer-2 B fragments are requested this many k-steps before their MFMAs
constexpr float kWeightTop = 16384.f;   // packer: largest |weight| of a layer after its power-of-two pre-scale
constexpr float kF16Max = 65504.f;
constexpr float kNegInf = -3.0e38f;     // max-pool padding on raw (pre-ReLU) accumulators
constexpr float kMean[3] = {0.485f, 0.456f, 0.406f}, kStd[3] = {0.229f, 0.224f, 0.225f};

// ---- LDS map (bytes) -------------------------------------------------------------------------------
constexpr int kMaxGroups = 4;                    // wave groups along M: 2 (4 waves, one per SIMD) or 4 (8 waves, two per SIMD)
constexpr int IN_PITCH = 72 * 8;                 // padded input row: 72 pixels x (R, G, B, mask) f16
constexpr int IN_BYTES = 70 * IN_PITCH;          // 3-pixel zero ring around 64x64
constexpr int IN8_PITCH = 72 * 4;                // int8 layer 1: 72 pixels x (R, G, B, mask) i8; pixel x sits at slot x + 4 (16-byte aligned quads), rows as above
constexpr int PX = 144;                          // pixel pitch of the f16 activation planes (128 B + 16 B skew)
constexpr int A2_PLANE = 257 * PX;               // 16x16 pixels + one all-zero pixel (index 256)
constexpr int A2H = IN_BYTES, A2L = A2H + A2_PLANE;
constexpr int A3_PLANE = 50 * PX;                // 7x7 pixels + one all-zero pixel (index 49)
constexpr int A3H = A2L + A2_PLANE, A3L = A3H + A3_PLANE;
constexpr int XCH = A3L + A3_PLANE;              // [groups-1][2][8][64] f32: first conv-2 row of group g+1, needed by group g's last pooled row
constexpr int FEAT = XCH + (kMaxGroups - 1) * 4096;   // [64] f32 pooled features
constexpr int RAW = FEAT + 256;                  // the NEXT frame's 12 288 raw bytes (prefetched during layer 2)
constexpr int LDS_TOTAL = RAW + kImg * kImg * 3;
constexpr int PART = A2H;                        // [groups-1][2][8][64] f32 layer-3 partial sums of K parts 1.. (A2 is dead by then)
static_assert(IN_BYTES % 16 == 0 && A2_PLANE % 16 == 0 && A3_PLANE % 16 == 0, "LDS planes must stay 16-byte aligned");
static_assert(LDS_TOTAL <= 160 * 1024, "encoder LDS map exceeds one CU");

constexpr size_t kPack1Bytes = 2 * kS1 * 64 * 32, kPack2Bytes = 2 * kS2 * 64 * 32;
constexpr size_t kPack1iBytes = 2 * kS1i * kDigits * 64 * 16;      // [n-half][k-step][digit][lane][16 i8]
constexpr int kDefaultGroups = 2;

struct EncParams {
    const uint8_t *images;      // [n][64][64][3]
    int n;
    const char *b1, *b2, *b3;   // packed B fragments: [n-half][k-step][lane][8 hi | 8 lo] f16
    const char *b1i;            // int8 layer 1: [n-half][k-step][digit][lane][16 i8]
    const float *inv1c;         // int8 layer 1: [64] per-output-channel 256 / fixed-point scale (powers of two)
    const float *inv_scale;     // [3] 1 / weight pre-scale of layers 1..3
    const float *bias2, *bias3, *fcw, *fcb;
    int state_dim;
    float *out;                 // [n][state_dim]
    int *status;                // bit 0: an activation left f16's range
    long long *prof;            // PROF only: [kProfFrames][kProfStamps][8 waves] cycle stamps of workgroup 0
};

extern __shared__ __attribute__((aligned(16))) char enc_lds[];

__device__ __forceinline__ f32x16 mfma16(half8 a, half8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ half8 lds16(int off) { return *reinterpret_cast<const half8 *>(enc_lds + off); }
// An MFMA the compiler may not move: `asm volatile` statements keep their order among themselves AND against every memory operation
// (ds_read / ds_write / global_load are chained behind unmodelled side effects), so a hand-written interleave of MFMAs and loads
// survives the scheduler exactly as written in the source — which it does not with the builtin: left to itself the scheduler issues
// the loads of a k-step in one burst between two MFMA groups (only ~5 single-issue instructions hide behind one 32-cycle MFMA with
// one wave per SIMD, MI355X_MICROARCH.md) and re-groups MFMAs into dependent runs on one accumulator.  The accumulator of a chain
// must be READ by compiler-visible code only after mfma_fence(): the hazard recogniser does not see the MFMA inside the statement.
// PAD = true puts two wait states in front of the MFMA, INSIDE the statement: the other hazard the compiler cannot see is a VALU write
// of an operand register directly before the statement (it reloads spilled fragments with v_accvgpr_read in the 256-register
// two-waves-per-SIMD instantiation; a reload issued right before its MFMA produced wrong features on the GPU, one instruction
// earlier it does not — profiles/NOTES.md section O).  The 512-register product has no such reloads (ISA lint, rule B) and runs unpadded.
template <bool PAD = false> __device__ __forceinline__ void mfma16_pinned(f32x16 &c, half8 a, half8 b) {
    if constexpr (PAD) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
// wait states between the last pinned MFMA of a phase and the first VALU / v_accvgpr_read of its result (8-pass XDL write -> VALU
// read needs 11 on gfx950; 20 issued)
__device__ __forceinline__ void mfma_fence() { asm volatile("s_nop 15\n\ts_nop 3"); }
// first MFMA of a chain: C = 0 as the inline constant (no 16 v_accvgpr_write per accumulator)
template <bool PAD = false> __device__ __forceinline__ void mfma16_pinned_first(f32x16 &c, half8 a, half8 b) {
    if constexpr (PAD) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=a"(c) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=a"(c) : "v"(a), "v"(b));
}
// "the results of these chains may be read from here on": an empty pinned statement that redefines the accumulators, placed (in source
// order = issue order) at least four pinned MFMAs (128 cycles) behind the last MFMA that wrote them — the compiler schedules every
// reader behind this statement, which keeps the XDL-write -> VALU-read distance the hazard recogniser cannot see
__device__ __forceinline__ void mfma_results_ready(f32x16 &a, f32x16 &b) { asm volatile("" : "+a"(a), "+a"(b)); }

// ---- int8 layer 1: pinned v_mfma_i32_32x32x32_i8.  A = 16 frame bytes per lane (VGPRs), B = 16 weight digits per lane (AGPRs: they
// stay resident for the whole layer), accumulators in VGPRs — the epilogue reads the sums without a v_accvgpr_read per value (an MFMA's
// vdst and srcC share one register bank: a chain cannot accumulate in AGPRs and deliver to VGPRs).
#if ENC_X == 6          // experiment: the chains accumulate in AGPRs (with ENC_X >= 4 nothing reads them)
#define ENC_ACC8 "a"
#else
#define ENC_ACC8 "v"
#endif
__device__ __forceinline__ void mfma8_first(i32x16 &c, i32x4 a, i32x4 b) {
    asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, 0" : "=&" ENC_ACC8(c) : "v"(a), "a"(b));
}
__device__ __forceinline__ void mfma8_next(i32x16 &c, i32x4 a, i32x4 b) {
    asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+" ENC_ACC8(c) : "v"(a), "a"(b));
}
// k-step `ks` of a chain (compile-time after unrolling)
__device__ __forceinline__ void mfma8_step(int ks, i32x16 &c, i32x4 a, i32x4 b) {
    if (ks == 0) mfma8_first(c, a, b);
    else mfma8_next(c, a, b);
}
// 16 frame bytes at an 8-byte aligned LDS offset (output pixel j starts at input pixel 2 j)
__device__ __forceinline__ i32x4 lds16u(int off) {
    const uint2 a = *reinterpret_cast<const uint2 *>(enc_lds + off), b = *reinterpret_cast<const uint2 *>(enc_lds + off + 8);
    i32x4 r; r[0] = (int)a.x; r[1] = (int)a.y; r[2] = (int)b.x; r[3] = (int)b.y;
    return r;
}
// the three digit sums of one output -> one int32 in units of 256 fixed-point steps: 256 a2 + a1 + floor(a0 / 256).  Exact but for the
// floor (< 1 unit in ~1e7); |result| <= 224 * 128 * kWeightTopI8 / 256 < 2^30.
__device__ __forceinline__ int comb3(int a2, int a1, int a0) { return a2 * 256 + a1 + (a0 >> 8); }
// the same as ONE pinned statement (ordered among the pinned MFMAs: plain VALU is not, the scheduler would issue the whole epilogue as
// one block instead of a few instructions per MFMA gap)
__device__ __forceinline__ int comb3_pinned(int a2, int a1, int a0) {
    int s, u;
    asm volatile("v_lshl_add_u32 %0, %2, 8, %3\n\tv_ashrrev_i32 %1, 8, %4\n\tv_add_u32 %0, %0, %1" : "=&v"(s), "=&v"(u) : "v"(a2), "v"(a1), "v"(a0));
    return s;
}
__device__ __forceinline__ int max3_pinned(int a, int b, int c) {
    int m;
    asm volatile("v_max3_i32 %0, %1, %2, %3" : "=v"(m) : "v"(a), "v"(b), "v"(c));
    return m;
}

// f32 -> (hi, lo) f16 planes at byte offset `off` inside the plane pair starting at hbase / lbase
__device__ __forceinline__ void store_split(int hbase, int lbase, int off, float v, bool &ovf) {
    ovf |= !(v < kF16Max);
    const _Float16 hi = (_Float16)v;
    const _Float16 lo = (_Float16)(v - (float)hi);
    *reinterpret_cast<_Float16 *>(enc_lds + hbase + off) = hi;
    *reinterpret_cast<_Float16 *>(enc_lds + lbase + off) = lo;
}

// two conv-1 output rows (32 pixels x this wave's 32 channels each): raw accumulators (pre-scaled weights).  Pinned MFMAs in issue
// order, the two accumulators alternating, the next k-step's fragments between them (see mfma16_pinned).
template <bool PAD> __device__ __forceinline__ void conv1_pair(const half8 (&Bh)[kS1], const half8 (&Bl)[kS1], int ra, int rb, int lane_base,
                                           f32x16 &v0, f32x16 &v1) {
    const int base0 = 2 * ra * IN_PITCH + lane_base, base1 = 2 * rb * IN_PITCH + lane_base;
    half8 x0 = lds16(base0), x1 = lds16(base1);
#pragma unroll
    for (int s = 0; s < kS1; s++) {
        const int offn = ((s + 1) >> 1) * IN_PITCH + ((s + 1) & 1) * 32;       // kernel row (s+1)/2, pixel slots 4((s+1)&1)+2h, +1
        half8 y0 = x0, y1 = x1;
        if (s == 0) mfma16_pinned_first<PAD>(v0, x0, Bh[0]); else mfma16_pinned<PAD>(v0, x0, Bh[s]);
        if (s + 1 < kS1) y0 = lds16(base0 + offn);
        if (s == 0) mfma16_pinned_first<PAD>(v1, x1, Bh[0]); else mfma16_pinned<PAD>(v1, x1, Bh[s]);
        if (s + 1 < kS1) y1 = lds16(base1 + offn);
        mfma16_pinned<PAD>(v0, x0, Bl[s]);
        mfma16_pinned<PAD>(v1, x1, Bl[s]);
        x0 = y0; x1 = y1;
    }
}
// one conv-1 output row: two chains (hi / lo weights), summed by the caller's reader
template <bool PAD> __device__ __forceinline__ void conv1_single(const half8 (&Bh)[kS1], const half8 (&Bl)[kS1], int ra, int lane_base, f32x16 &v0) {
    f32x16 a0, a1;
    const int base0 = 2 * ra * IN_PITCH + lane_base;
    half8 x0 = lds16(base0);
#pragma unroll
    for (int s = 0; s < kS1; s++) {
        half8 y0 = x0;
        if (s == 0) mfma16_pinned_first<PAD>(a0, x0, Bh[0]); else mfma16_pinned<PAD>(a0, x0, Bh[s]);
        if (s + 1 < kS1) y0 = lds16(base0 + ((s + 1) >> 1) * IN_PITCH + ((s + 1) & 1) * 32);
        if (s == 0) mfma16_pinned_first<PAD>(a1, x0, Bl[0]); else mfma16_pinned<PAD>(a1, x0, Bl[s]);
        x0 = y0;
    }
    asm volatile("s_nop 15\n\ts_nop 3" : "+a"(a0), "+a"(a1));              // XDL write -> VALU read distance (see mfma_fence)
#pragma unroll
    for (int r = 0; r < 16; r++) v0[r] = a0[r] + a1[r];
}

// int8 layer 1, not pipelined: one conv row -> its three digit sums (21 pinned MFMAs, the three chains rotating)
__device__ __forceinline__ void conv1_row_i8(const i32x4 (&Bd)[kDigits][kS1i], int ra, int lane_base, i32x16 (&R)[kDigits]) {
    const int base0 = 2 * ra * IN8_PITCH + lane_base;
    i32x4 x0 = lds16u(base0);
#pragma unroll
    for (int ks = 0; ks < kS1i; ks++) {
        i32x4 y0 = x0;
#pragma unroll
        for (int d = 0; d < kDigits; d++) {
            mfma8_step(ks, R[d], x0, Bd[d][ks]);
            if (d == 0 && ks + 1 < kS1i) y0 = lds16u(base0 + (ks + 1) * IN8_PITCH);
        }
        x0 = y0;
    }
}
// one conv row -> combined int32 sums
__device__ __forceinline__ void conv1_single_i8(const i32x4 (&Bd)[kDigits][kS1i], int ra, int lane_base, i32x16 &out) {
    i32x16 R[kDigits];
    conv1_row_i8(Bd, ra, lane_base, R);
    asm volatile("s_nop 15\n\ts_nop 3" : "+v"(R[0]), "+v"(R[1]), "+v"(R[2]));      // XDL write -> VALU read distance (see mfma_fence)
#pragma unroll
    for (int r = 0; r < 16; r++) out[r] = comb3(R[0][r], R[1][r], R[2][r]);
}

// PROF: workgroup 0 stamps s_memtime at 9 points of its first kProfFrames frames (srlhip_encoder_phase_cycles)
constexpr int kProfFrames = 8, kProfStamps = 9, kProfWaves = 2 * kMaxGroups;
#define ENC_STAMP(k)                                                                                          \
    do {                                                                                                      \
        if (PROF && blockIdx.x == 0 && lane == 0 && frame < kProfFrames)                                      \
            P.prof[(frame * kProfStamps + (k)) * kProfWaves + wave] = (long long)__builtin_readcyclecounter(); \
    } while (0)

// MG = wave groups along M (output pixels): group g owns 16/MG pooled layer-1 rows, 8/MG layer-2 tiles, 36/MG of
// layer 3's k-steps; the two waves of a group own 32 output channels each.
// I8 = layer 1 on the int8 matrix pipe (the product; MG = 2 only), else the split-f16 layer 1 of rounds 2-5.
template <bool PROF, int MG, bool I8>
__global__ __launch_bounds__(128 * MG, 1) void encoder_fwd_k(EncParams P) {
    static_assert(!I8 || MG == 2, "the int8 layer 1 is written for one wave per SIMD");
    constexpr int kThreads = 128 * MG, kRows1 = 16 / MG, kTiles2 = 8 / MG, kSteps3 = kS2 / MG;
    constexpr int kRawIters = (768 + kThreads - 1) / kThreads;          // 16-byte pieces of a frame per thread
    const int tid0 = threadIdx.x;
    int tid = tid0, wave = tid >> 6, lane = tid & 63;
    int mh = wave >> 1, nh = wave & 1;            // which wave group (part of the pixels) / half of the output channels this wave owns
    int j = lane & 31, h = lane >> 5;             // MFMA lane coordinates: column (channel) j, k / row half h
    int ch = 32 * nh + j;                         // the output channel this lane owns in every layer
    bool ovf = false;

    // -- once per workgroup: zero the padded input (ring stays zero), the all-zero pixels; stage the first frame
    for (int o = tid * 16; o < IN_BYTES; o += kThreads * 16) *reinterpret_cast<uint4 *>(enc_lds + o) = make_uint4(0, 0, 0, 0);
    if (tid < 9) {
        *reinterpret_cast<uint4 *>(enc_lds + A2H + 256 * PX + tid * 16) = make_uint4(0, 0, 0, 0);
        *reinterpret_cast<uint4 *>(enc_lds + A2L + 256 * PX + tid * 16) = make_uint4(0, 0, 0, 0);
        *reinterpret_cast<uint4 *>(enc_lds + A3H + 49 * PX + tid * 16) = make_uint4(0, 0, 0, 0);
        *reinterpret_cast<uint4 *>(enc_lds + A3L + 49 * PX + tid * 16) = make_uint4(0, 0, 0, 0);
    }
    const float bias2 = P.bias2[ch], bias3 = P.bias3[ch];
    const float inv1 = I8 ? P.inv1c[ch] : P.inv_scale[0], inv2 = P.inv_scale[1], inv3 = P.inv_scale[2];
    if (blockIdx.x < P.n) {                       // first frame of this workgroup (later ones are prefetched in layer 2)
        const uint4 *src = reinterpret_cast<const uint4 *>(P.images + (size_t)blockIdx.x * (kImg * kImg * 3));
        uint4 *raw = reinterpret_cast<uint4 *>(enc_lds + RAW);
#pragma unroll
        for (int i = 0; i < kRawIters; i++)
            if (tid + i * kThreads < 768) raw[tid + i * kThreads] = src[tid + i * kThreads];
    }
    __syncthreads();

    // int8 layer 1: this wave's weight digits, 84 AGPRs, loaded ONCE per workgroup and resident across frames (the AGPR half of the
    // register file is otherwise idle outside the accumulators; the f16 form reloads its 112 VGPRs of fragments every frame).  The
    // pinned statement puts them in their AGPRs here: a v_accvgpr_write the compiler would otherwise sink in front of its first pinned
    // MFMA is the VALU-write -> MFMA-operand hazard the recogniser cannot see inside the statement (ISA lint rule B).
    i32x4 Bd[kDigits][kS1i];
    if constexpr (I8) {
#pragma unroll
        for (int s = 0; s < kS1i; s++)
#pragma unroll
            for (int d = 0; d < kDigits; d++)
                Bd[d][s] = *reinterpret_cast<const i32x4 *>(P.b1i + ((size_t)((nh * kS1i + s) * kDigits + d) * 64 + lane) * 16);
#pragma unroll
        for (int d = 0; d < kDigits; d++)
            asm volatile("s_nop 4" : "+a"(Bd[d][0]), "+a"(Bd[d][1]), "+a"(Bd[d][2]), "+a"(Bd[d][3]), "+a"(Bd[d][4]), "+a"(Bd[d][5]), "+a"(Bd[d][6]));
    }

    // layer 3's A-fragment offsets (lane constants with a division each): computed once per workgroup, explicitly — the int8 variant
    // re-derives the lane coordinates per frame (below) and would otherwise recompute them between layer 3's pinned MFMAs
    int a3o[kSteps3];
    {
        const int oy = (j >> 2) & 3, ox = j & 3;
#pragma unroll
        for (int ii = 0; ii < kSteps3; ii++) {
            const int s = kSteps3 * mh + ii, tap = s >> 2, q = s & 3;
            const int ky = tap / 3, kx = tap - 3 * ky;
            const int sy = 2 * oy + ky - 1, sx = 2 * ox + kx - 1;
            const bool ok = j < 16 && (unsigned)sy < 7u && (unsigned)sx < 7u;
            a3o[ii] = (ok ? sy * 7 + sx : 49) * PX + h * 16 + q * 32;
        }
    }

    int frame = 0;
    for (int img = blockIdx.x; img < P.n; img += gridDim.x, frame++) {
        if constexpr (I8) {
            // The lane coordinates are re-derived per frame from an opaque copy of the thread id: everything computed from them (dozens of
            // LDS / global addresses of layers 2, 3 and the FC) is then recomputed where it is used instead of being hoisted out of the
            // frame loop and kept live across it — with the digits resident in AGPRs those hoisted values were spilled to scratch and
            // reloaded inside the layer-2 / layer-3 epilogues (+5 k cycles per frame).
            tid = tid0;
            asm volatile("" : "+v"(tid));
            wave = tid >> 6; lane = tid & 63; mh = wave >> 1; nh = wave & 1; j = lane & 31; h = lane >> 5; ch = 32 * nh + j;
        }
        ENC_STAMP(0);
        if constexpr (I8) {
        // ---- int8 layer 1 -------------------------------------------------------------------------------------------------------------
        // ---- phase 0: uint8 frame (staged in LDS) -> (R - 128, G - 128, B - 128, 127) int8 pixels inside the zero ring: p ^ 0x80 per byte;
        // the validity mask is 127, not 1, so that its tap (weight / 127) is quantised on the same scale as the colour taps (|p - 128| <= 128)
        uint32_t rawq[1024 / kThreads][3];
#pragma unroll
        for (int i = 0; i < 1024 / kThreads; i++) {
            const uint32_t *w = reinterpret_cast<const uint32_t *>(enc_lds + RAW + 12 * (tid + kThreads * i));
            rawq[i][0] = w[0]; rawq[i][1] = w[1]; rawq[i][2] = w[2];
        }
#pragma unroll
        for (int i = 0; i < 1024 / kThreads; i++) {
            const int q = tid + kThreads * i;                              // four consecutive pixels = 12 bytes
            const uint32_t w0 = rawq[i][0], w1 = rawq[i][1], w2 = rawq[i][2];
            const uint32_t px[4] = {w0 & 0xffffffu, (w0 >> 24) | ((w1 & 0xffffu) << 8), (w1 >> 16) | ((w2 & 0xffu) << 16), w2 >> 8};
            const int y = (4 * q) >> 6, x = (4 * q) & 63;
            *reinterpret_cast<uint4 *>(enc_lds + ((y + 3) * 72 + (x + 4)) * 4) =
                make_uint4((px[0] ^ 0x808080u) | kMaskByte, (px[1] ^ 0x808080u) | kMaskByte, (px[2] ^ 0x808080u) | kMaskByte, (px[3] ^ 0x808080u) | kMaskByte);
        }
        __syncthreads();
        ENC_STAMP(1);
        // wave group g owns pooled rows R g .. R g + R-1 <- conv rows 2R g - 1 .. 2R (g+1) - 1 (its first row is recomputed, not exchanged).
        // Output pixel j of a conv row reads input pixels 2 j - 3 .. 2 j + 3 = slots 2 j + 1 .. 2 j + 7: the fragment starts at slot 2 j
        // (8-byte aligned), its first slot carries zero weights.
        {
            const int lane_base = j * 8 + h * 16;
            i32x16 carry;
            i32x16 R0[kDigits], R1[kDigits];
            if (mh == 0) {
#pragma unroll
                for (int r = 0; r < 16; r++) carry[r] = kIntNegInf;        // conv row -1 is pool padding
            } else {
                conv1_single_i8(Bd, 2 * kRows1 * mh - 1, lane_base, carry);
            }
            conv1_row_i8(Bd, 2 * kRows1 * mh, lane_base, R0);
            conv1_row_i8(Bd, 2 * kRows1 * mh + 1, lane_base, R1);
            // One pipeline step = csrc/encoder_l1_step.inc (GENERATED by gen/gen_encoder_l1_step.py: every MFMA gap holds at most seven
            // other instructions).  N0 <- conv row 2 prow + 2 (slots 0..20), N1 <- conv row 2 prow + 3 (slots 21..41): 21 pinned MFMAs
            // each, k-step major, the three digit chains rotating.  In their shadow: the A fragments two k-steps ahead (three rotating
            // registers F0..F2; the last two loads of a step are the next step's first fragments), pooled row prow <- (carry, S0, Q1)
            // — S0 = row 0 of the previous pair, already recombined; Q1 = row 1's three digit sums — and this pair's row-0 recombination
            // into S0.  carry <- Q1 recombined.  Every VALU piece is a pinned statement (plain VALU is not ordered against the MFMAs).
            i32x16 S0;
            asm volatile("s_nop 15\n\ts_nop 3" : "+v"(R0[0]), "+v"(R0[1]), "+v"(R0[2]));
#pragma unroll
            for (int r = 0; r < 16; r++) S0[r] = comb3(R0[0][r], R0[1][r], R0[2][r]);
            const unsigned long long hmask = 0xffffffff00000000ull;          // lanes 32..63: h = 1
            int mx = 0;                                                      // largest pooled value of the layer (overflow watch)
            auto step = [&](int prow, i32x16 (&Q1)[kDigits], i32x16 (&N1)[kDigits], i32x4 &F0, i32x4 &F1, i32x4 &F2) {
                const int base0 = 2 * (2 * prow + 2) * IN8_PITCH + lane_base;
                i32x16 N0[kDigits];
                i32x16 m;
                int neg = kIntNegInf, left[4], ka[4], kb[4];
                float pf[4][2];
                unsigned phi[4][2];
                int baseH = 0, baseL = 0;
#define L1_MF(row, ks, d, fv) mfma8_step(ks, (row) ? N1[d] : N0[d], F##fv, Bd[d][ks])
#define L1_LD(n, fv) F##fv = lds16u(base0 + (((n) / kS1i) * 2 + (n) % kS1i) * IN8_PITCH)
#define L1_BASE() do { baseH = A2H + (prow * 16 + 2 * h) * PX + ch * 2; baseL = baseH + A2_PLANE; } while (0)
#define L1_READY() asm volatile("" : "+v"(Q1[0]), "+v"(Q1[1]), "+v"(Q1[2]))
                // row 1 of the previous pair, register r: digits -> one int32; 3-row max; the row becomes the next pair's carry
#define L1_C1(r) do { int s1_, m_;                                                                                          \
        asm volatile("v_lshl_add_u32 %0, %2, 8, %3\n\tv_ashrrev_i32 %1, 8, %4\n\tv_add_u32 %0, %0, %1\n\tv_max3_i32 %1, %5, %6, %0" \
                     : "=&v"(s1_), "=&v"(m_) : "v"(Q1[0][r]), "v"(Q1[1][r]), "v"(Q1[2][r]), "v"(carry[r]), "v"(S0[r]));              \
        carry[r] = s1_; m[r] = m_; } while (0)
                // row 0 of THIS pair, register r (its chains finished at slot 20)
#define L1_C0(r) do { int s0_, u_;                                                                                          \
        asm volatile("v_lshl_add_u32 %0, %2, 8, %3\n\tv_ashrrev_i32 %1, 8, %4\n\tv_add_u32 %0, %0, %1"                       \
                     : "=&v"(s0_), "=&v"(u_) : "v"(N0[0][r]), "v"(N0[1][r]), "v"(N0[2][r]));                                  \
        S0[r] = s0_; } while (0)
                // pooled pixel 2 h + 1 of group g: window registers 4 g + 1 .. 4 g + 3, ReLU in integers
#define L1_KB(g) asm volatile("v_max3_i32 %0, %1, %2, %3\n\tv_max_i32 %0, 0, %0" : "=&v"(kb[g]) : "v"(m[4 * (g) + 1]), "v"(m[4 * (g) + 2]), "v"(m[4 * (g) + 3]))
                // the pixel left of a lane's 4-group is lane^32's: h = 1 lanes need the partner's register 4 g + 3, h = 0 lanes its register
                // 4 g - 1 (nothing for g = 0).  v_permlane32_swap A, B exchanges A's upper with B's lower half-wave: A = m[4 g - 1], B = m[4 g + 3],
                // in place (KB(g), KB(g - 1) have read them; the half of B the next group needs is the one the swap leaves alone)
#define L1_XC(g) do { int a_ = (g) ? m[(g) ? 4 * (g) - 1 : 0] : neg, b_ = m[4 * (g) + 3];                                 \
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_cndmask_b32_e64 %2, %1, %0, %3"                             \
                     : "+v"(a_), "+v"(b_), "=&v"(left[g]) : "s"(hmask));                                                       \
        m[4 * (g) + 3] = b_; } while (0)
#define L1_KA(g) asm volatile("v_max3_i32 %0, %1, %2, %3\n\tv_max_i32 %0, 0, %0" : "=&v"(ka[g]) : "v"(m[4 * (g)]), "v"(m[4 * (g) + 1]), "v"(left[g]))
#define L1_MX(g) asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(mx) : "v"(ka[g]), "v"(kb[g]))
                // pooled pixel (g, u): int -> f32 scale -> f16 hi -> A2H; then lo = f16(f - hi) -> A2L
#define L1_PB(g, u) do { asm volatile("v_cvt_f32_i32 %0, %2\n\tv_mul_f32 %0, %3, %0\n\tv_cvt_f16_f32 %1, %0"               \
                                      : "=&v"(pf[g][u]), "=&v"(phi[g][u]) : "v"((u) ? kb[g] : ka[g]), "v"(inv1));              \
        *reinterpret_cast<unsigned short *>(enc_lds + baseH + (4 * (g) + (u)) * PX) = (unsigned short)phi[g][u]; } while (0)
#define L1_PC(g, u) do { unsigned lo_;                                                                                      \
        asm volatile("v_cvt_f32_f16 %0, %1\n\tv_sub_f32 %0, %2, %0\n\tv_cvt_f16_f32 %0, %0" : "=&v"(lo_) : "v"(phi[g][u]), "v"(pf[g][u])); \
        *reinterpret_cast<unsigned short *>(enc_lds + baseL + (4 * (g) + (u)) * PX) = (unsigned short)lo_; } while (0)
#if ENC_X >= 4          // experiments 4-6: the bare MFMA + fragment-load stream of the step (results are wrong by construction)
#undef L1_READY
#undef L1_C1
#undef L1_C0
#undef L1_KB
#undef L1_XC
#undef L1_KA
#undef L1_MX
#undef L1_PB
#undef L1_PC
#define L1_READY() (void)0
#define L1_C1(r) (void)0
#define L1_C0(r) (void)0
#define L1_KB(g) (void)0
#define L1_XC(g) (void)0
#define L1_KA(g) (void)0
#define L1_MX(g) (void)0
#define L1_PB(g, u) (void)0
#define L1_PC(g, u) (void)0
#endif
#if ENC_X == 5          // ... without the fragment loads
#undef L1_LD
#define L1_LD(n, fv) (void)0
#endif
#include "encoder_l1_step.inc"
#if ENC_X >= 4
                asm volatile("" :: ENC_ACC8(N0[0]), ENC_ACC8(N0[1]), ENC_ACC8(N0[2]), ENC_ACC8(N1[0]), ENC_ACC8(N1[1]), ENC_ACC8(N1[2]));
#endif
#undef L1_MF
#undef L1_LD
#undef L1_BASE
#undef L1_READY
#undef L1_C1
#undef L1_C0
#undef L1_KB
#undef L1_XC
#undef L1_KA
#undef L1_MX
#undef L1_PB
#undef L1_PC
            };
            static_assert((kRows1 - 1) % 2 == 1, "the pipeline below is written for an odd number of hidden pooled rows");
            i32x16 W1[kDigits];
            int prow = kRows1 * mh;
            // the first step's first two fragments (conv row 2 prow + 2, k-steps 0 and 1)
            i32x4 F0 = lds16u(2 * (2 * prow + 2) * IN8_PITCH + lane_base), F1 = lds16u(2 * (2 * prow + 2) * IN8_PITCH + lane_base + IN8_PITCH), F2 = F0;
#pragma unroll 1
            for (int pp = 0; pp < (kRows1 - 1) / 2; pp++, prow += 2) {
                step(prow, R1, W1, F0, F1, F2);
                step(prow + 1, W1, R1, F2, F0, F1);        // a step leaves the next one's fragments in its third and first register
                F0 = F1; F1 = F2;
            }
            step(prow, R1, W1, F0, F1, F2);
            // the last pooled row of this group: nothing left to hide it behind
            asm volatile("s_nop 15\n\ts_nop 3" : "+v"(W1[0]), "+v"(W1[1]), "+v"(W1[2]));
            {
                const int last = kRows1 * mh + kRows1 - 1;
                i32x16 m;
#pragma unroll
                for (int r = 0; r < 16; r++) m[r] = max(carry[r], max(S0[r], comb3(W1[0][r], W1[1][r], W1[2][r])));
                const int t3 = __shfl_xor(m[3], 32), t7 = __shfl_xor(m[7], 32), t11 = __shfl_xor(m[11], 32), t15 = __shfl_xor(m[15], 32);
                const int left[4] = {h ? t3 : kIntNegInf, h ? t7 : t3, h ? t11 : t7, h ? t15 : t11};
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const int ka = max(max(max(m[4 * g], m[4 * g + 1]), left[g]), 0);
                    const int kb = max(max(max(m[4 * g + 1], m[4 * g + 2]), m[4 * g + 3]), 0);
                    mx = max(mx, max(ka, kb));
                    const int k = 4 * g + 2 * h;
                    bool dummy = false;
                    store_split(A2H, A2L, (last * 16 + k) * PX + ch * 2, (float)ka * inv1, dummy);
                    store_split(A2H, A2L, (last * 16 + k + 1) * PX + ch * 2, (float)kb * inv1, dummy);
                }
            }
            ovf |= !((float)mx * inv1 < kF16Max);
        }
        } else {
        // this wave's layer-1 B fragments (112 VGPRs, in registers for the whole layer): requested before the unpack.
        // (Keeping them — or layer 3's — resident across frames starves the layer-2 loop of registers: its prefetch
        // ring then becomes loop-carried copies behind `s_waitcnt vmcnt(0)` and spills; measured slower.)
        half8 B1h[kS1], B1l[kS1];
#pragma unroll
        for (int s = 0; s < kS1; s++) {
            const half8 *p = reinterpret_cast<const half8 *>(P.b1 + ((size_t)(nh * kS1 + s) * 64 + lane) * 32);
            B1h[s] = p[0];
            B1l[s] = p[1];
        }
        // ---- phase 0: uint8 frame (already staged in LDS) -> f16 (R, G, B, 1) pixels inside the zero ring ------
        // (every raw word is read before the first pixel is written: the compiler cannot prove that the IN stores do not alias the RAW
        //  reads, so a read-convert-write loop waits out one LDS round trip per piece)
        uint32_t rawq[1024 / kThreads][3];
#pragma unroll
        for (int i = 0; i < 1024 / kThreads; i++) {
            const uint32_t *w = reinterpret_cast<const uint32_t *>(enc_lds + RAW + 12 * (tid + kThreads * i));
            rawq[i][0] = w[0]; rawq[i][1] = w[1]; rawq[i][2] = w[2];
        }
#pragma unroll
        for (int i = 0; i < 1024 / kThreads; i++) {
            const int q = tid + kThreads * i;                              // four consecutive pixels = 12 bytes
            const uint32_t w0 = rawq[i][0], w1 = rawq[i][1], w2 = rawq[i][2];
            const uint32_t px[4] = {w0 & 0xffffffu, (w0 >> 24) | ((w1 & 0xffffu) << 8), (w1 >> 16) | ((w2 & 0xffu) << 16), w2 >> 8};
            const int y = (4 * q) >> 6, x = (4 * q) & 63;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                half4v v;
                v[0] = (_Float16)(float)(px[k] & 0xffu);
                v[1] = (_Float16)(float)((px[k] >> 8) & 0xffu);
                v[2] = (_Float16)(float)((px[k] >> 16) & 0xffu);
                v[3] = (_Float16)1.0f;
                *reinterpret_cast<half4v *>(enc_lds + ((y + 3) * 72 + (x + k + 3)) * 8) = v;
            }
        }
        __syncthreads();
        ENC_STAMP(1);

        // ---- layer 1: conv7x7/2 + ReLU + maxpool3/2 p1 -> A2 planes (16x16x64 f16 hi/lo) ----------------------
        // wave group g owns pooled rows R g .. R g + R-1 (R = 16 / MG) <- conv rows 2R g - 1 .. 2R (g+1) - 1 (its first
        // row is recomputed, not exchanged)
        {
            const int lane_base = j * 16 + h * 16;
            f32x16 carry, v0, v1;
            if (mh == 0) {
#pragma unroll
                for (int r = 0; r < 16; r++) carry[r] = kNegInf;           // conv row -1 is pool padding
            } else {
                conv1_single<MG != 2>(B1h, B1l, 2 * kRows1 * mh - 1, lane_base, carry);
            }
            // Software pipeline: the 56 MFMAs of pooled row p+1 and the pooling / split / store epilogue of row p are
            // independent instruction streams in one basic block, so the epilogue's VALU and LDS work issues in the
            // shadow of the matrix pipe (one wave per SIMD: nothing else would hide it).
            auto epilogue = [&](int prow, const f32x16 &above, const f32x16 &r0, const f32x16 &r1) {
                f32x16 m;
#pragma unroll
                for (int r = 0; r < 16; r++) m[r] = fmaxf(above[r], fmaxf(r0[r], r1[r]));
                // a lane holds pixels x = 8g + 4h + r (register 4g + r); the pixel left of its 4-group is lane^32's
                const float t3 = __shfl_xor(m[3], 32), t7 = __shfl_xor(m[7], 32), t11 = __shfl_xor(m[11], 32),
                            t15 = __shfl_xor(m[15], 32);
                const float left[4] = {h ? t3 : kNegInf, h ? t7 : t3, h ? t11 : t7, h ? t15 : t11};
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const float ka = fmaxf(fmaxf(m[4 * g], m[4 * g + 1]), left[g]);
                    const float kb = fmaxf(fmaxf(m[4 * g + 1], m[4 * g + 2]), m[4 * g + 3]);
                    const int k = 4 * g + 2 * h;
                    store_split(A2H, A2L, (prow * 16 + k) * PX + ch * 2, fmaxf(0.f, ka * inv1), ovf);
                    store_split(A2H, A2L, (prow * 16 + k + 1) * PX + ch * 2, fmaxf(0.f, kb * inv1), ovf);
                }
            };
            if constexpr (MG == 2) {
                conv1_pair<MG != 2>(B1h, B1l, 2 * kRows1 * mh, 2 * kRows1 * mh + 1, lane_base, v0, v1);
                // Software pipeline, written out by hand (round 5): the 56 MFMAs of the NEXT two conv rows are pinned statements in
                // issue order — the two accumulators alternate, so no MFMA waits for its predecessor — and between them, in source
                // order, sit the next k-step's two A fragments and the sixteen LDS stores of the PREVIOUS pair's pooled row (its max /
                // scale / split arithmetic is plain VALU that the scheduler places in front of the store that needs it).  Fully unrolled:
                // carry / v0 / v1 rotate by renaming (the rolled loop spent ~250 cycles per pooled row on 56 register moves), and the
                // sched_group_barrier pipeline it replaces let the scheduler re-group the MFMAs into runs of 6-8 on ONE accumulator.
                // One pipeline step: (n0, n1) <- conv rows 2 prow + 2, 2 prow + 3; pooled row prow <- (carry, v0, v1); carry <- v1.
                auto pipe_step = [&](int prow, f32x16 &v0_, f32x16 &v1_, f32x16 &n0, f32x16 &n1) {
                    const int base0 = 2 * (2 * prow + 2) * IN_PITCH + lane_base, base1 = 2 * (2 * prow + 3) * IN_PITCH + lane_base;
                    half8 x0 = lds16(base0), x1 = lds16(base1), y0 = x0, y1 = x1;
                    f32x16 m;
                    float left[4] = {0.f, 0.f, 0.f, 0.f};
                    // MFMA i of the step (k-step i / 4: n0 * hi, n1 * hi, n0 * lo, n1 * lo) and, behind it in issue order, slot i of
                    // the previous pooled row's epilogue.  Only ~5 single-issue instructions hide behind one MFMA, so the epilogue is
                    // dealt out a few instructions per slot; its accumulator reads are pinned statements too (plain VALU is not ordered
                    // against the pinned MFMAs: the scheduler would issue all ~90 instructions of the max stage in one block).
#pragma unroll
                    for (int i = 0; i < 4 * kS1; i++) {
                        const int ks = i >> 2, q = i & 3;
                        if (i == 0) mfma16_pinned_first(n0, x0, B1h[0]);
                        else if (i == 1) mfma16_pinned_first(n1, x1, B1h[0]);
                        else if (q == 0) mfma16_pinned(n0, x0, B1h[ks]);
                        else if (q == 1) mfma16_pinned(n1, x1, B1h[ks]);
                        else if (q == 2) mfma16_pinned(n0, x0, B1l[ks]);
                        else mfma16_pinned(n1, x1, B1l[ks]);
                        if (ks + 1 < kS1) {                                 // the next k-step's fragments
                            const int offn = ((ks + 1) >> 1) * IN_PITCH + ((ks + 1) & 1) * 32;
                            if (q == 0) y0 = lds16(base0 + offn);
                            if (q == 1) y1 = lds16(base1 + offn);
                        }
                        if (q == 3) { x0 = y0; x1 = y1; }
                        if (i == 4) mfma_results_ready(v0_, v1_);           // last written >= 5 pinned MFMAs (160 cycles) ago
                        if (i >= 5 && i < 21) {                             // max over the three conv rows, one register per slot
                            const int r = i - 5;
                            float e0, e1;
                            asm volatile("v_accvgpr_read_b32 %0, %2\n\tv_accvgpr_read_b32 %1, %3" : "=&v"(e0), "=v"(e1) : "a"(v0_[r]), "a"(v1_[r]));
                            m[r] = fmaxf(carry[r], fmaxf(e0, e1));
                            carry[r] = e1;
                        }
                        if (i == 21) {
                            const float t3 = __shfl_xor(m[3], 32), t7 = __shfl_xor(m[7], 32), t11 = __shfl_xor(m[11], 32),
                                        t15 = __shfl_xor(m[15], 32);
                            left[0] = h ? t3 : kNegInf; left[1] = h ? t7 : t3; left[2] = h ? t11 : t7; left[3] = h ? t15 : t11;
                        }
                        if (i >= 24 && i < 48 && (i - 24) % 3 == 0) {       // eight pooled pixels, one split store pair per third slot
                            const int u = (i - 24) / 3, g = u >> 1;
                            const float kv = (u & 1) ? fmaxf(fmaxf(m[4 * g + 1], m[4 * g + 2]), m[4 * g + 3])
                                                     : fmaxf(fmaxf(m[4 * g], m[4 * g + 1]), left[g]);
                            store_split(A2H, A2L, (prow * 16 + 4 * g + 2 * h + (u & 1)) * PX + ch * 2, fmaxf(0.f, kv * inv1), ovf);
                        }
                    }
                };
                // two steps per rolled iteration: the accumulator pairs (v0, v1) and (w0, w1) swap roles without a register move
                static_assert((kRows1 - 1) % 2 == 1, "the pipeline below is written for an odd number of hidden pooled rows");
                f32x16 w0, w1;
                int prow = kRows1 * mh;
#pragma unroll 1
                for (int pp = 0; pp < (kRows1 - 1) / 2; pp++, prow += 2) {
                    pipe_step(prow, v0, v1, w0, w1);
                    pipe_step(prow + 1, w0, w1, v0, v1);
                }
                pipe_step(prow, v0, v1, w0, w1);
                v0 = w0; v1 = w1;
                // the last pooled row of this group: nothing left to hide it behind
                asm volatile("s_nop 15\n\ts_nop 3" : "+a"(v0), "+a"(v1));
                epilogue(kRows1 * mh + kRows1 - 1, carry, v0, v1);
            } else {
                // two waves per SIMD: the partner wave's MFMAs cover this wave's epilogue, no in-wave pipelining (registers)
#pragma unroll 1
                for (int p = 0; p < kRows1; p++) {
                    const int prow = kRows1 * mh + p;
                    conv1_pair<MG != 2>(B1h, B1l, 2 * prow, 2 * prow + 1, lane_base, v0, v1);
                    asm volatile("s_nop 15\n\ts_nop 3" : "+a"(v0), "+a"(v1));      // pinned MFMAs: XDL write -> VALU read distance
                    epilogue(prow, carry, v0, v1);
                    carry = v1;
                }
            }
        }
        }
        ENC_STAMP(2);
        __syncthreads();
        ENC_STAMP(3);

        // ---- layer 2: conv3x3 p1 + ReLU + maxpool3/2 -> A3 planes (7x7x64) -----------------------------------
        // wave group g owns tiles T = K g .. K g + K-1 (K = 8 / MG; tile = conv rows 2T, 2T+1 x 16 columns) -> pooled rows T
        half8 B3h[kSteps3], B3l[kSteps3];
        {
            f32x16 acc[kTiles2];
#pragma unroll
            for (int t = 0; t < kTiles2; t++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[t][r] = 0.f;
            const int row = j >> 4, ox = j & 15;
            const char *bp = P.b2 + ((size_t)(nh * kS2) * 64 + lane) * 32;
            // the next frame of this workgroup: fetched now, parked in LDS after the k-loop
            const int next_img = img + gridDim.x;
            uint4 nxt_raw[kRawIters];
#pragma unroll
            for (int i = 0; i < kRawIters; i++) nxt_raw[i] = make_uint4(0, 0, 0, 0);
            if (next_img < P.n) {
                const uint4 *src = reinterpret_cast<const uint4 *>(P.images + (size_t)next_img * (kImg * kImg * 3));
#pragma unroll
                for (int i = 0; i < kRawIters; i++)
                    if (tid + i * kThreads < 768) nxt_raw[i] = src[tid + i * kThreads];
            }
            // Layer-2 B fragments come from L2 (they do not fit in the registers the resident layers leave) through a
            // ring of kB2Ahead register slots: the fragment of k-step s + kB2Ahead is requested right after step s has
            // used its slot (~2300 MFMA cycles ahead; an L2 hit costs ~1600 cycles here — two steps ahead left the loop
            // latency-bound at half the MFMA rate).
            half8 Rh[kB2Ahead], Rl[kB2Ahead];
#pragma unroll
            for (int s = 0; s < kB2Ahead; s++) {
                const half8 *pn = reinterpret_cast<const half8 *>(bp + (size_t)s * 64 * 32);
                Rh[s] = pn[0];
                Rl[s] = pn[1];
            }
            // A fragments are double-buffered by hand (the loads of step s+1 are issued before the MFMAs of step s)
            // and the MFMA order is pinned with scheduling barriers: left alone, the compiler serialises
            // ds_read -> wait -> two dependent MFMAs on one accumulator, which runs at half the matrix rate.
            int addr[kTiles2];
            auto tap_addr = [&](int ky, int kx) {
#pragma unroll
                for (int t = 0; t < kTiles2; t++) {
                    const int sy = 2 * (kTiles2 * mh + t) + row + ky - 1, sx = ox + kx - 1;
                    const bool ok = (unsigned)sy < 16u && (unsigned)sx < 16u;
                    addr[t] = (ok ? sy * 16 + sx : 256) * PX + h * 16;
                }
            };
            half8 ah[kTiles2], al[kTiles2], nh_[kTiles2], nl_[kTiles2];
            tap_addr(0, 0);
#pragma unroll
            for (int t = 0; t < kTiles2; t++) { ah[t] = lds16(A2H + addr[t]); al[t] = lds16(A2L + addr[t]); }
            // One rolled iteration = one kernel row = 12 k-steps, so ring slots are compile-time indices.  A k-step is 3 * kTiles2
            // pinned MFMAs (hi*hi, lo*hi, hi*lo per tile: consecutive MFMAs never share an accumulator) with the step's other work —
            // the next step's 2 * kTiles2 A fragments, the refill of the ring slot — written BETWEEN them, one load per MFMA gap.
#pragma unroll 1
            for (int ky = 0; ky < 3; ky++) {
#pragma unroll
                for (int i = 0; i < 12; i++) {
                    const int slot = i % kB2Ahead;
                    const int in = (i + 1) % 12, qn = in & 3;
                    if (qn == 0) tap_addr(in == 0 ? ky + 1 : ky, in >> 2);   // ky + 1 == 3: every row is out of range -> zero pixel
                    const int sn = 12 * ky + i + kB2Ahead;
                    const half8 *pn = reinterpret_cast<const half8 *>(bp + (size_t)(sn < kS2 ? sn : kS2 - 1) * 64 * 32);
                    const half8 bh = Rh[slot], bl = Rl[slot];
#if ENC_X == 1 || ENC_X == 3            // experiment: no LDS reads in the loop
#pragma unroll
                    for (int t = 0; t < kTiles2; t++) { mfma16_pinned<MG != 2>(acc[t], ah[t], bh); nh_[t] = ah[t]; }
#pragma unroll
                    for (int t = 0; t < kTiles2; t++) { mfma16_pinned<MG != 2>(acc[t], al[t], bh); nl_[t] = al[t]; }
#else
#pragma unroll
                    for (int t = 0; t < kTiles2; t++) { mfma16_pinned<MG != 2>(acc[t], ah[t], bh); nh_[t] = lds16(A2H + addr[t] + qn * 32); }
#pragma unroll
                    for (int t = 0; t < kTiles2; t++) { mfma16_pinned<MG != 2>(acc[t], al[t], bh); nl_[t] = lds16(A2L + addr[t] + qn * 32); }
#endif
#if ENC_X != 2 && ENC_X != 3            // experiment 2: no B streaming (3: neither LDS reads nor B streaming)
                    Rh[slot] = pn[0];                                      // (the last requests of a frame re-read fragment 35: harmless)
#endif
#pragma unroll
                    for (int t = 0; t < kTiles2; t++) mfma16_pinned<MG != 2>(acc[t], ah[t], bl);
#if ENC_X != 2 && ENC_X != 3
                    Rl[slot] = pn[1];
#endif
#pragma unroll
                    for (int t = 0; t < kTiles2; t++) { ah[t] = nh_[t]; al[t] = nl_[t]; }
                }
            }
            // (tied to the accumulators: their readers are scheduled behind it)
            if constexpr (kTiles2 == 4) asm volatile("s_nop 15\n\ts_nop 3" : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]), "+a"(acc[3]));
            else { asm volatile("s_nop 15\n\ts_nop 3" : "+a"(acc[0]), "+a"(acc[1])); }
            ENC_STAMP(4);
            // layer 3's B fragments for this wave's K part: one burst, in flight during the epilogue below
#pragma unroll
            for (int ii = 0; ii < kSteps3; ii++) {
                const half8 *p3 = reinterpret_cast<const half8 *>(P.b3 + ((size_t)(nh * kS2 + kSteps3 * mh + ii) * 64 + lane) * 32);
                B3h[ii] = p3[0];
                B3l[ii] = p3[1];
            }
            if (next_img < P.n) {
                uint4 *raw = reinterpret_cast<uint4 *>(enc_lds + RAW);
#pragma unroll
                for (int i = 0; i < kRawIters; i++)
                    if (tid + i * kThreads < 768) raw[tid + i * kThreads] = nxt_raw[i];
            }
            // registers 0..7 = conv row 2T, 8..15 = row 2T+1; pixel x = 4h + (r & 3) + 8 ((r >> 2) & 1)
            if (mh > 0) {                                                  // this group's first conv row completes the previous group's last pooled row
                float *xw = reinterpret_cast<float *>(enc_lds + XCH) + ((mh - 1) * 2 + nh) * 8 * 64 + lane;
#pragma unroll
                for (int r = 0; r < 8; r++) xw[r * 64] = acc[0][r];
            }
            __syncthreads();
            const float *xch = reinterpret_cast<const float *>(enc_lds + XCH) + ((mh < MG - 1 ? mh : MG - 2) * 2 + nh) * 8 * 64 + lane;
        ENC_STAMP(5);
#pragma unroll
            for (int t = 0; t < kTiles2; t++) {
                if (t == kTiles2 - 1 && mh == MG - 1) break;               // pooled row 7 does not exist
                float w[8];
#pragma unroll
                for (int r = 0; r < 8; r++) {
                    const float below = t < kTiles2 - 1 ? acc[t < kTiles2 - 1 ? t + 1 : kTiles2 - 1][r] : xch[r * 64];
                    w[r] = fmaxf(fmaxf(acc[t][r], acc[t][r + 8]), below);
                }
                const float t0 = __shfl_xor(w[0], 32), t4 = __shfl_xor(w[4], 32);
                const int prow = kTiles2 * mh + t;
                const float ka = fmaxf(fmaxf(w[0], w[1]), w[2]);           // x = 4h .. 4h+2      -> k = 2h
                const float kb = fmaxf(fmaxf(w[2], w[3]), h ? t4 : t0);    // x = 4h+2 .. 4h+4    -> k = 2h+1
                const float kc = fmaxf(fmaxf(w[4], w[5]), w[6]);           // x = 8+4h .. 8+4h+2  -> k = 4+2h
                store_split(A3H, A3L, (prow * 7 + 2 * h) * PX + ch * 2, fmaxf(0.f, fmaf(ka, inv2, bias2)), ovf);
                store_split(A3H, A3L, (prow * 7 + 2 * h + 1) * PX + ch * 2, fmaxf(0.f, fmaf(kb, inv2, bias2)), ovf);
                store_split(A3H, A3L, (prow * 7 + 4 + 2 * h) * PX + ch * 2, fmaxf(0.f, fmaf(kc, inv2, bias2)), ovf);
                if (h == 0) {
                    const float kd = fmaxf(fmaxf(w[6], w[7]), t4);         // x = 10, 11, 12      -> k = 5
                    store_split(A3H, A3L, (prow * 7 + 5) * PX + ch * 2, fmaxf(0.f, fmaf(kd, inv2, bias2)), ovf);
                }
            }
        }
        __syncthreads();
        ENC_STAMP(6);

        // ---- layer 3: conv3x3/2 p1 (7x7 -> 4x4) + ReLU + maxpool3/2 (-> 1x1); K split over the wave groups ----
        {
            f32x16 c0;
            auto a3_addr = [&](int ii) { return a3o[ii]; };
            {
                // three chains (hi*hi, hi*lo, lo*hi), pinned in issue order so that no MFMA waits for its predecessor; the next k-step's
                // two A fragments are requested between them (the compiler's schedule read them right before their use and chained the
                // two MFMAs of one accumulator back to back): 4.1 k -> 2.9 k cycles for the phase
                f32x16 c1, c2;
                half8 ah = lds16(A3H + a3_addr(0)), al = lds16(A3L + a3_addr(0));
#pragma unroll
                for (int ii = 0; ii < kSteps3; ii++) {
                    half8 nh2 = ah, nl2 = al;
                    if (ii == 0) mfma16_pinned_first<MG != 2>(c0, ah, B3h[0]); else mfma16_pinned<MG != 2>(c0, ah, B3h[ii]);
                    if (ii + 1 < kSteps3) nh2 = lds16(A3H + a3_addr(ii + 1));
                    if (ii == 0) mfma16_pinned_first<MG != 2>(c1, ah, B3l[0]); else mfma16_pinned<MG != 2>(c1, ah, B3l[ii]);
                    if (ii + 1 < kSteps3) nl2 = lds16(A3L + a3_addr(ii + 1));
                    if (ii == 0) mfma16_pinned_first<MG != 2>(c2, al, B3h[0]); else mfma16_pinned<MG != 2>(c2, al, B3h[ii]);
                    ah = nh2; al = nl2;
                }
                asm volatile("s_nop 15\n\ts_nop 3" : "+a"(c0), "+a"(c1), "+a"(c2));       // XDL write -> VALU read distance
#pragma unroll
                for (int r = 0; r < 16; r++) c0[r] += c1[r] + c2[r];
            }
            if (mh > 0) {                                                  // A2 is dead: its first KiBs carry the partial sums
                float *pw = reinterpret_cast<float *>(enc_lds + PART) + ((mh - 1) * 2 + nh) * 8 * 64 + lane;
#pragma unroll
                for (int r = 0; r < 8; r++) pw[r * 64] = c0[r];
            }
            __syncthreads();
            const float *part = reinterpret_cast<const float *>(enc_lds + PART) + nh * 8 * 64 + lane;
            if (mh == 0) {
                // registers 0..3 -> output pixels 4h + r, 4..7 -> 8 + 4h + (r - 4); the 3x3 pool window is
                // pixels {0,1,2,4,5,6,8,9,10}
                float m = kNegInf;
#pragma unroll
                for (int r = 0; r < 8; r++) {
                    float v = c0[r];
#pragma unroll
                    for (int g = 0; g < MG - 1; g++) v += part[(g * 16 + r) * 64];
                    const bool in_window = (r & 3) != 3 && (h == 0 || r < 4);
                    if (in_window) m = fmaxf(m, v);
                }
                m = fmaxf(m, __shfl_xor(m, 32));
                if (h == 0) reinterpret_cast<float *>(enc_lds + FEAT)[ch] = fmaxf(0.f, fmaf(m, inv3, bias3));
            }
        }
        __syncthreads();
        ENC_STAMP(7);

        // ---- FC: state = fc_w . features + fc_b ----------------------------------------------------------------
        {
            // four lanes per output component (16 features each), reduced with two shuffles
            const float *feat = reinterpret_cast<const float *>(enc_lds + FEAT);
            const int part4 = tid & 3;
            for (int sd0 = 0; sd0 < P.state_dim; sd0 += kThreads / 4) {
                const int sd = sd0 + (tid >> 2);
                float a = 0.f;
                if (sd < P.state_dim) {
                    const float4 *wr = reinterpret_cast<const float4 *>(P.fcw + (size_t)sd * kCh + 16 * part4);
                    const float4 *fr = reinterpret_cast<const float4 *>(feat + 16 * part4);
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        const float4 wv = wr[c], fv = fr[c];
                        a = fmaf(wv.x, fv.x, a); a = fmaf(wv.y, fv.y, a); a = fmaf(wv.z, fv.z, a); a = fmaf(wv.w, fv.w, a);
                    }
                }
                a += __shfl_xor(a, 1);
                a += __shfl_xor(a, 2);
                if (sd < P.state_dim && part4 == 0) P.out[(size_t)img * P.state_dim + sd] = a + P.fcb[sd];
            }
        }
        ENC_STAMP(8);
        // The next writers of FEAT / XCH / A3 / IN sit behind the barriers of the next frame; its layer 1 overwrites
        // A2 (= PART) only after the barrier that follows its unpack, which every wave reaches after reading PART.
    }
    if (ovf) atomicOr(P.status, 1);
}
#undef ENC_STAMP

// ---------------------------------------------------------------------------------------------- host side
using srlenc::pack_layer3x3;
using srlenc::pick_scale;
using srlenc::split_f16;
double layer1_weight(const float *w, const float *b, int o, int k) {
    const int ky = k / 32, kx = (k % 32) / 4, c4 = k % 4;
    double v = 0.0;
    if (kx < 7) {
        if (c4 < 3) {
            v = (double)w[((o * 3 + c4) * 7 + kx) * 7 + ky] / (255.0 * (double)kStd[c4]);
        } else {
            for (int c = 0; c < 3; c++) v -= (double)w[((o * 3 + c) * 7 + kx) * 7 + ky] * (double)kMean[c] / (double)kStd[c];
            if (ky == 3 && kx == 3) v += (double)b[o];
        }
    }
    return v;
}

// conv1_w [64][3][7][7] (torch OIHW, BatchNorm folded), conv1_b [64].  The network sees the frame with its two
// spatial axes swapped (models.py:185-188), so the tap at frame offset (ky, kx) is torch's w[o][c][kx][ky].
float pack_layer1(const float *w, const float *b, _Float16 *out) {
    double wmax = 0.0;
    for (int o = 0; o < 64; o++)
        for (int k = 0; k < 16 * kS1; k++) wmax = fmax(wmax, fabs(layer1_weight(w, b, o, k)));
    const float scale = pick_scale(wmax);
    for (int nh = 0; nh < 2; nh++)
        for (int s = 0; s < kS1; s++)
            for (int lane = 0; lane < 64; lane++)
                for (int e = 0; e < 8; e++) {
                    const int o = 32 * nh + (lane & 31), k = 16 * s + 8 * (lane >> 5) + e;
                    _Float16 *dst = out + ((size_t)(nh * kS1 + s) * 64 + lane) * 16;
                    split_f16((float)(layer1_weight(w, b, o, k) * (double)scale), dst[e], dst[8 + e]);
                }
    return scale;
}
}  // namespace

namespace srlenc {
// int8 layer 1 (3-channel frames): the folded taps around p - 128.  k = ky * 32 + slot * 4 + c4; ONE of the eight pixel slots of a kernel
// row carries zero weights — slot 0 in the fused kernel (the fragment of output pixel j starts at input pixel 2 j - 4, 16-byte aligned
// quads), slot 7 in the layered kernels (the fragment starts at the window column of tap 0); c4 < 3: w / (255 std_c) (multiplies
// p - 128); c4 = 3, the validity mask (byte value 127 inside the frame, 0 outside): [sum_c w_c (128 / 255 - mean_c) / std_c — zero
// padding stays exact in NORMALISED space — plus the folded BN bias on the centre tap] / 127.
static double layer1_weight_i8(const float *w, const float *b, int zero_slot, int o, int k) {
    constexpr float kMean[3] = {0.485f, 0.456f, 0.406f}, kStd[3] = {0.229f, 0.224f, 0.225f};
    const int ky = k / 32, slot = (k % 32) / 4, c4 = k % 4, kx = zero_slot == 0 ? slot - 1 : slot;
    double v = 0.0;
    if (slot != zero_slot) {
        if (c4 < 3) {
            v = (double)w[((o * 3 + c4) * 7 + kx) * 7 + ky] / (255.0 * (double)kStd[c4]);
        } else {
            for (int c = 0; c < 3; c++) v += (double)w[((o * 3 + c) * 7 + kx) * 7 + ky] * (128.0 / 255.0 - (double)kMean[c]) / (double)kStd[c];
            if (ky == 3 && kx == 3) v += (double)b[o];
            v /= (double)kMaskI8;
        }
    }
    return v;
}
// out: [n-half][k-step][digit][lane][16 i8], digit 0 = most significant; inv256[o] = 256 / scale_o, scale_o = the largest power of two
// with max_k |w| scale_o <= kWeightTopI8: weight = (65536 d0 + 256 d1 + d2) / scale_o with balanced digits in [-128, 127].
void pack_layer1_i8(const float *w, const float *b, int zero_slot, int8_t *out, float *inv256) {
    for (int o = 0; o < 64; o++) {
        double wmax = 0.0;
        for (int k = 0; k < 32 * kI8Steps; k++) wmax = fmax(wmax, fabs(layer1_weight_i8(w, b, zero_slot, o, k)));
        int e = 0;
        if (wmax > 0.0 && std::isfinite(wmax)) {
            frexp((double)kWeightTopI8 / wmax, &e);      // kWeightTopI8 / wmax = f * 2^e, f in [0.5, 1)
            e = e - 1 > 60 ? 60 : (e - 1 < -60 ? -60 : e - 1);
        }
        const double scale = ldexp(1.0, e);
        inv256[o] = (float)(256.0 / scale);
        const int nh = o >> 5;
        for (int k = 0; k < 32 * kI8Steps; k++) {
            long long z = llround(layer1_weight_i8(w, b, zero_slot, o, k) * scale);
            if (z > kWeightTopI8) z = kWeightTopI8;
            if (z < -kWeightTopI8) z = -kWeightTopI8;
            int dg[kI8Digits];
            for (int d = kI8Digits - 1; d >= 0; d--) {          // least significant first
                long long r = ((z + 128) % 256 + 256) % 256 - 128;
                dg[d] = (int)r;
                z = (z - r) / 256;
            }
            const int s = k / 32, hh = (k % 32) / 16, el = k % 16, lane = hh * 32 + (o & 31);
            for (int d = 0; d < kI8Digits; d++)
                out[((size_t)((nh * kI8Steps + s) * kI8Digits + d) * 64 + lane) * 16 + el] = (int8_t)dg[d];
        }
    }
}

void split_f16(float v, _Float16 &hi, _Float16 &lo) {
    hi = (_Float16)v;
    lo = (_Float16)(v - (float)hi);
}
// largest power of two that keeps max|w| * scale <= kWeightTop (lo parts then sit well inside f16's normal range)
float pick_scale(double wmax) {
    if (!(wmax > 0.0) || !std::isfinite(wmax)) return 1.f;
    int e;
    frexp((double)kWeightTop / wmax, &e);          // kWeightTop / wmax = f * 2^e, f in [0.5, 1)
    e = e - 1 > 40 ? 40 : (e - 1 < -40 ? -40 : e - 1);
    return (float)ldexp(1.0, e);
}
// conv_w [64][64][3][3] (torch OIHW, BatchNorm folded): k = (ky * 3 + kx) * 64 + c
float pack_layer3x3(const float *w, _Float16 *out) {
    double wmax = 0.0;
    for (int i = 0; i < 64 * 64 * 9; i++) wmax = fmax(wmax, fabs((double)w[i]));
    const float scale = pick_scale(wmax);
    for (int nh = 0; nh < 2; nh++)
        for (int s = 0; s < kS2; s++)
            for (int lane = 0; lane < 64; lane++)
                for (int e = 0; e < 8; e++) {
                    const int o = 32 * nh + (lane & 31), k = 16 * s + 8 * (lane >> 5) + e;
                    const int tap = k / 64, c = k % 64, ky = tap / 3, kx = tap % 3;
                    _Float16 *dst = out + ((size_t)(nh * kS2 + s) * 64 + lane) * 16;
                    split_f16(w[((o * 64 + c) * 3 + kx) * 3 + ky] * scale, dst[e], dst[8 + e]);
                }
    return scale;
}

}  // namespace srlenc

struct srlhip_encoder {
    int device_id, state_dim;
    char *d_pack;          // b1 | b2 | b3
    float *d_f32;          // inv_scale[3] pad bias2[64] bias3[64] fcw[state_dim][64] fcb[state_dim]
    int *d_status;
    int num_cus;
    int groups;            // wave groups along M: 2 = 4 waves per workgroup, 4 = 8 waves
    int l1_i8;             // layer 1 on the int8 matrix pipe (default with groups == 2); SRLHIP_ENCODER_L1=f16 keeps the split-f16 form
    char *d_pack_i8;       // int8 layer-1 digits
    float *d_inv1c;        // [64] per-channel 256 / scale
    srlenc::General *general;   // non-null: the layered path of encoder_general.hip serves this handle (any shape but 64x64x3)
    int img_h, img_w, n_channels;
    std::string err;
    int fail(int code, const std::string &m) { err = m; return code; }
};

namespace {
thread_local std::string g_enc_create_error;

bool fused_shape(int img_h, int img_w, int n_channels) {
    const char *g = getenv("SRLHIP_ENCODER_GENERAL");          // test knob: run 64x64x3 through the layered path too
    return img_h == kImg && img_w == kImg && n_channels == 3 && !(g && atoi(g) == 1);
}
bool supported_shape(int img_h, int img_w, int n_channels) { return srlenc::geometry(img_h, img_w, n_channels).ok; }
}  // namespace

extern "C" {

size_t srlhip_encoder_pack_bytes(void) { return kPack1Bytes + 2 * kPack2Bytes; }

size_t srlhip_encoder_pack_i8_bytes(void) { return kPack1iBytes; }

int srlhip_encoder_pack_i8(const float *conv1_w, const float *conv1_b, int32_t zero_slot, void *out, size_t out_bytes, float *inv_scale64) {
    if (!conv1_w || !conv1_b || !out || !inv_scale64 || out_bytes < kPack1iBytes || (zero_slot != 0 && zero_slot != 7)) return SRLHIP_EINVAL;
    srlenc::pack_layer1_i8(conv1_w, conv1_b, zero_slot, static_cast<int8_t *>(out), inv_scale64);
    return SRLHIP_OK;
}

int srlhip_encoder_pack(const float *conv1_w, const float *conv1_b, const float *conv2_w, const float *conv3_w,
                        void *out, size_t out_bytes, float *scales3) {
    if (!conv1_w || !conv1_b || !conv2_w || !conv3_w || !out || !scales3 || out_bytes < srlhip_encoder_pack_bytes())
        return SRLHIP_EINVAL;
    char *p = static_cast<char *>(out);
    scales3[0] = pack_layer1(conv1_w, conv1_b, reinterpret_cast<_Float16 *>(p));
    scales3[1] = pack_layer3x3(conv2_w, reinterpret_cast<_Float16 *>(p + kPack1Bytes));
    scales3[2] = pack_layer3x3(conv3_w, reinterpret_cast<_Float16 *>(p + kPack1Bytes + kPack2Bytes));
    return SRLHIP_OK;
}

int srlhip_encoder_pack_first_layer(int32_t n_channels, const float *conv1_w, const float *conv1_b, void *out, size_t out_bytes, float *scale) {
    if (!scale) return SRLHIP_EINVAL;
    *scale = srlenc::pack_layer1_general(conv1_w, conv1_b, n_channels, out, out_bytes);
    return *scale > 0.f ? SRLHIP_OK : SRLHIP_EINVAL;
}

int srlhip_encoder_supported(int32_t img_h, int32_t img_w, int32_t n_channels) { return supported_shape(img_h, img_w, n_channels) ? 1 : 0; }

int32_t srlhip_encoder_feature_count(int32_t img_h, int32_t img_w, int32_t n_channels) {
    const srlenc::Geometry g = srlenc::geometry(img_h, img_w, n_channels);
    return g.ok ? 64 * g.Hp[2] * g.Wp[2] : 0;
}

int srlhip_encoder_create(int32_t device_id, int32_t img_h, int32_t img_w, int32_t n_channels, int32_t state_dim,
                          const float *conv1_w, const float *conv1_b, const float *conv2_w, const float *conv2_b,
                          const float *conv3_w, const float *conv3_b, const float *fc_w, const float *fc_b,
                          srlhip_encoder_handle *out) {
    if (!out) return SRLHIP_EINVAL;
    *out = nullptr;
    if (!supported_shape(img_h, img_w, n_channels) || state_dim < 1 || !conv2_b || !conv3_b || !fc_w || !fc_b) {
        g_enc_create_error = "srlhip_encoder_create: CustomCNN frames are 8..1024 pixels a side with 3 or 6 channels, state_dim >= 1";
        return SRLHIP_ENOTSUP;
    }
    if (!fused_shape(img_h, img_w, n_channels)) {
        srlhip_encoder *e = new (std::nothrow) srlhip_encoder();
        if (!e) return SRLHIP_ENOMEM;
        e->device_id = device_id; e->state_dim = state_dim; e->d_pack = nullptr; e->d_f32 = nullptr; e->d_status = nullptr; e->general = nullptr;
        e->img_h = img_h; e->img_w = img_w; e->n_channels = n_channels; e->d_pack_i8 = nullptr; e->d_inv1c = nullptr; e->l1_i8 = 0;
        const int rc = srlenc::general_create(device_id, srlenc::geometry(img_h, img_w, n_channels), state_dim, conv1_w, conv1_b, conv2_w, conv2_b,
                                              conv3_w, conv3_b, fc_w, fc_b, &e->general, g_enc_create_error);
        if (rc != SRLHIP_OK) { delete e; return rc; }
        *out = e;
        return SRLHIP_OK;
    }
    std::vector<char> pack(srlhip_encoder_pack_bytes());
    float scales[3];
    if (srlhip_encoder_pack(conv1_w, conv1_b, conv2_w, conv3_w, pack.data(), pack.size(), scales) != SRLHIP_OK) {
        g_enc_create_error = "srlhip_encoder_create: null weight pointer";
        return SRLHIP_EINVAL;
    }
    srlhip_encoder *e = new (std::nothrow) srlhip_encoder();
    if (!e) return SRLHIP_ENOMEM;
    e->device_id = device_id; e->state_dim = state_dim; e->d_pack = nullptr; e->d_f32 = nullptr; e->d_status = nullptr; e->general = nullptr;
    e->img_h = img_h; e->img_w = img_w; e->n_channels = n_channels; e->d_pack_i8 = nullptr; e->d_inv1c = nullptr; e->l1_i8 = 0;
    std::vector<int8_t> pack_i8(kPack1iBytes);
    float inv1c[64];
    srlenc::pack_layer1_i8(conv1_w, conv1_b, 0, pack_i8.data(), inv1c);
#define ENC_CHECK(expr)                                                                              \
    do {                                                                                             \
        hipError_t e__ = (expr);                                                                     \
        if (e__ != hipSuccess) {                                                                     \
            g_enc_create_error = std::string(#expr ": ") + hipGetErrorString(e__);                   \
            srlhip_encoder_destroy(e);                                                               \
            return SRLHIP_EHIP;                                                                      \
        }                                                                                            \
    } while (0)
    ENC_CHECK(hipSetDevice(device_id));
    hipDeviceProp_t prop;
    ENC_CHECK(hipGetDeviceProperties(&prop, device_id));
    e->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    const size_t nf = 132 + (size_t)state_dim * (1 + kCh);
    std::vector<float> f(nf);
    for (int i = 0; i < 3; i++) f[i] = 1.f / scales[i];      // powers of two: exact
    f[3] = 0.f;
    memcpy(f.data() + 4, conv2_b, 64 * sizeof(float));
    memcpy(f.data() + 68, conv3_b, 64 * sizeof(float));
    memcpy(f.data() + 132, fc_w, (size_t)state_dim * kCh * sizeof(float));        // 16-byte aligned rows (float4 loads)
    memcpy(f.data() + 132 + (size_t)state_dim * kCh, fc_b, state_dim * sizeof(float));
    ENC_CHECK(hipMalloc(reinterpret_cast<void **>(&e->d_pack), pack.size()));
    ENC_CHECK(hipMalloc(reinterpret_cast<void **>(&e->d_f32), nf * sizeof(float)));
    ENC_CHECK(hipMalloc(reinterpret_cast<void **>(&e->d_status), sizeof(int)));
    ENC_CHECK(hipMemcpy(e->d_pack, pack.data(), pack.size(), hipMemcpyHostToDevice));
    ENC_CHECK(hipMemcpy(e->d_f32, f.data(), nf * sizeof(float), hipMemcpyHostToDevice));
    ENC_CHECK(hipMemset(e->d_status, 0, sizeof(int)));
    ENC_CHECK(hipMalloc(reinterpret_cast<void **>(&e->d_pack_i8), kPack1iBytes));
    ENC_CHECK(hipMalloc(reinterpret_cast<void **>(&e->d_inv1c), sizeof inv1c));
    ENC_CHECK(hipMemcpy(e->d_pack_i8, pack_i8.data(), kPack1iBytes, hipMemcpyHostToDevice));
    ENC_CHECK(hipMemcpy(e->d_inv1c, inv1c, sizeof inv1c, hipMemcpyHostToDevice));
    ENC_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(encoder_fwd_k<false, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL));
    ENC_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(encoder_fwd_k<true, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL));
    ENC_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(encoder_fwd_k<false, 2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL));
    ENC_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(encoder_fwd_k<true, 2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL));
    ENC_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(encoder_fwd_k<false, 4, false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL));
    ENC_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(encoder_fwd_k<true, 4, false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL));
    {
        const char *w = getenv("SRLHIP_ENCODER_WAVES");       // experiment knob: 4 (one wave per SIMD) or 8 (two)
        e->groups = (w && atoi(w) == 4) ? 2 : (w && atoi(w) == 8) ? 4 : kDefaultGroups;
        const char *l1 = getenv("SRLHIP_ENCODER_L1");         // experiment knob: f16 = the split-f16 layer 1 of rounds 2-5
        e->l1_i8 = e->groups == 2 && !(l1 && strcmp(l1, "f16") == 0);
    }
#undef ENC_CHECK
    *out = e;
    return SRLHIP_OK;
}

int srlhip_encoder_forward(srlhip_encoder_handle e, const uint8_t *images_dev, int32_t n, float *states_dev, void *hip_stream) {
    if (!e) return SRLHIP_EINVAL;
    if (n < 0 || (n > 0 && (!images_dev || !states_dev))) return e->fail(SRLHIP_EINVAL, "srlhip_encoder_forward: null buffer");
    if (n == 0) return SRLHIP_OK;
    if (reinterpret_cast<uintptr_t>(images_dev) % 16) return e->fail(SRLHIP_EINVAL, "srlhip_encoder_forward: images must be 16-byte aligned");
    hipError_t rc = hipSetDevice(e->device_id);
    if (rc != hipSuccess) return e->fail(SRLHIP_EHIP, std::string("hipSetDevice: ") + hipGetErrorString(rc));
    if (e->general) return srlenc::general_forward(e->general, images_dev, n, states_dev, static_cast<hipStream_t>(hip_stream), e->err);
    EncParams p;
    p.images = images_dev; p.n = n;
    p.b1 = e->d_pack; p.b2 = e->d_pack + kPack1Bytes; p.b3 = e->d_pack + kPack1Bytes + kPack2Bytes;
    p.inv_scale = e->d_f32; p.bias2 = e->d_f32 + 4; p.bias3 = e->d_f32 + 68; p.fcw = e->d_f32 + 132;
    p.fcb = e->d_f32 + 132 + (size_t)e->state_dim * kCh;
    p.state_dim = e->state_dim; p.out = states_dev; p.status = e->d_status; p.prof = nullptr;
    p.b1i = e->d_pack_i8; p.inv1c = e->d_inv1c;
    const int grid = n < e->num_cus ? n : e->num_cus;
    if (e->groups == 2 && e->l1_i8) hipLaunchKernelGGL((encoder_fwd_k<false, 2, true>), dim3(grid), dim3(256), LDS_TOTAL, static_cast<hipStream_t>(hip_stream), p);
    else if (e->groups == 2) hipLaunchKernelGGL((encoder_fwd_k<false, 2, false>), dim3(grid), dim3(256), LDS_TOTAL, static_cast<hipStream_t>(hip_stream), p);
    else hipLaunchKernelGGL((encoder_fwd_k<false, 4, false>), dim3(grid), dim3(512), LDS_TOTAL, static_cast<hipStream_t>(hip_stream), p);
    rc = hipGetLastError();
    if (rc != hipSuccess) return e->fail(SRLHIP_EHIP, std::string("encoder_fwd_k launch: ") + hipGetErrorString(rc));
    return SRLHIP_OK;
}

int srlhip_encoder_phase_cycles(srlhip_encoder_handle e, const uint8_t *images_dev, int32_t n, float *states_dev,
                                int64_t *cycles9) {
    if (!e || !cycles9) return SRLHIP_EINVAL;
    if (n < 1 || !images_dev || !states_dev) return e->fail(SRLHIP_EINVAL, "srlhip_encoder_phase_cycles: need n >= 1 and both buffers");
    if (e->general) return e->fail(SRLHIP_ENOTSUP, "srlhip_encoder_phase_cycles: the phase stamps belong to the fused 64x64x3 kernel");
#define ENC_RC(expr)                                                                                  \
    do {                                                                                              \
        hipError_t e__ = (expr);                                                                      \
        if (e__ != hipSuccess) return e->fail(SRLHIP_EHIP, std::string(#expr ": ") + hipGetErrorString(e__)); \
    } while (0)
    ENC_RC(hipSetDevice(e->device_id));
    const size_t words = (size_t)kProfFrames * kProfStamps * kProfWaves;
    long long *d_prof = nullptr;
    ENC_RC(hipMalloc(reinterpret_cast<void **>(&d_prof), words * sizeof(long long)));
    ENC_RC(hipMemset(d_prof, 0, words * sizeof(long long)));
    EncParams p;
    p.images = images_dev; p.n = n;
    p.b1 = e->d_pack; p.b2 = e->d_pack + kPack1Bytes; p.b3 = e->d_pack + kPack1Bytes + kPack2Bytes;
    p.inv_scale = e->d_f32; p.bias2 = e->d_f32 + 4; p.bias3 = e->d_f32 + 68; p.fcw = e->d_f32 + 132;
    p.fcb = e->d_f32 + 132 + (size_t)e->state_dim * kCh;
    p.state_dim = e->state_dim; p.out = states_dev; p.status = e->d_status; p.prof = d_prof;
    p.b1i = e->d_pack_i8; p.inv1c = e->d_inv1c;
    const int grid = n < e->num_cus ? n : e->num_cus;
    if (e->groups == 2 && e->l1_i8) hipLaunchKernelGGL((encoder_fwd_k<true, 2, true>), dim3(grid), dim3(256), LDS_TOTAL, nullptr, p);
    else if (e->groups == 2) hipLaunchKernelGGL((encoder_fwd_k<true, 2, false>), dim3(grid), dim3(256), LDS_TOTAL, nullptr, p);
    else hipLaunchKernelGGL((encoder_fwd_k<true, 4, false>), dim3(grid), dim3(512), LDS_TOTAL, nullptr, p);
    ENC_RC(hipGetLastError());
    std::vector<long long> host(words);
    ENC_RC(hipMemcpy(host.data(), d_prof, words * sizeof(long long), hipMemcpyDeviceToHost));
    (void)hipFree(d_prof);
#undef ENC_RC
    // segment k = stamp k -> stamp k+1 (the last one wraps to the next frame's stamp 0), slowest wave, averaged over
    // the frames workgroup 0 processed
    const int frames = (n + grid - 1) / grid < kProfFrames ? (n - 1) / grid + 1 : kProfFrames;
    for (int k = 0; k < kProfStamps; k++) cycles9[k] = 0;
    int counted = 0;
    for (int f = 0; f + 1 < frames || (frames == 1 && f == 0); f++) {
        for (int k = 0; k < kProfStamps; k++) {
            long long worst = 0;
            for (int w = 0; w < 2 * e->groups; w++) {
                const long long a = host[(f * kProfStamps + k) * kProfWaves + w];
                const bool last = k + 1 == kProfStamps;
                if (last && f + 1 >= frames) continue;
                const long long b = last ? host[((f + 1) * kProfStamps) * kProfWaves + w] : host[(f * kProfStamps + k + 1) * kProfWaves + w];
                if (b - a > worst) worst = b - a;
            }
            cycles9[k] += worst;
        }
        counted++;
    }
    if (counted > 1)
        for (int k = 0; k < kProfStamps; k++) cycles9[k] /= counted;
    return SRLHIP_OK;
}

int srlhip_encoder_overflow(srlhip_encoder_handle e, int32_t *flag) {
    if (!e || !flag) return SRLHIP_EINVAL;
    hipError_t rc = hipSetDevice(e->device_id);
    if (rc == hipSuccess) rc = hipDeviceSynchronize();
    int v = 0;
    if (rc == hipSuccess) rc = hipMemcpy(&v, e->general ? srlenc::general_status(e->general) : e->d_status, sizeof(int), hipMemcpyDeviceToHost);
    if (rc != hipSuccess) return e->fail(SRLHIP_EHIP, std::string("srlhip_encoder_overflow: ") + hipGetErrorString(rc));
    *flag = v & 1;
    return SRLHIP_OK;
}

int srlhip_encoder_destroy(srlhip_encoder_handle e) {
    if (!e) return SRLHIP_OK;
    (void)hipSetDevice(e->device_id);
    if (e->general) srlenc::general_destroy(e->general);
    if (e->d_pack) (void)hipFree(e->d_pack);
    if (e->d_pack_i8) (void)hipFree(e->d_pack_i8);
    if (e->d_inv1c) (void)hipFree(e->d_inv1c);
    if (e->d_f32) (void)hipFree(e->d_f32);
    if (e->d_status) (void)hipFree(e->d_status);
    delete e;
    return SRLHIP_OK;
}

const char *srlhip_encoder_last_error(srlhip_encoder_handle e) { return e ? e->err.c_str() : g_enc_create_error.c_str(); }

}  // extern "C"
