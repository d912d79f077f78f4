// encoder.hip — fused SRL encoder forward (srl_zoo CustomCNN) for the raw_pixels path on gfx950.
//
// Replaces, for a whole device-resident batch of rasterised frames:
//   SRLNeuralNetwork.getState        state_representation/models.py:178-193  (preprocess + model forward)
//   MultiprocessSRLModel._run        rl_baselines/utils.py:181-191           (one image at a time behind queues)
// CustomCNN (restated from srl_zoo; BatchNorms folded by the host side):
//   conv7x7/2 p3 (3->64) + ReLU + maxpool3/2 p1 -> conv3x3 p1 (64->64) + ReLU + maxpool3/2
//   -> conv3x3/2 p1 (64->64) + ReLU + maxpool3/2 -> FC(64 -> state_dim)            at 64x64x3 input.
//
// MI355X design: ONE kernel, one 256-lane workgroup per CU looping over images; an image never leaves
// the CU between the uint8 frame (12 KiB read from HBM) and its state vector (state_dim floats written):
//   * every activation lives in LDS (136 KiB of the CU's 160 KiB); the three convolutions are implicit GEMMs on
//     v_mfma_f32_32x32x16_f16 with A = pixels (im2col fragments are contiguous 16-byte LDS reads: channels are
//     the fastest index, the 3-channel input is repacked to RGB+mask so a pixel is 8 bytes) and B = weights;
//   * f32 accuracy on the f16 matrix pipe (16x the f32 MFMA rate): every operand is split x = hi + lo / 2048
//     with hi = f16(x), lo = f16((x - hi) * 2048); products hi*hi go to one accumulator, hi*lo + lo*hi to a
//     second one that is folded in at the end (lo*lo ~ 2^-22 is dropped).  uint8 pixels are exact in f16, so
//     layer 1 needs only the weight split (2 MFMAs), layers 2-3 need 3;
//   * the ImageNet normalisation of preprocessImage is folded into the layer-1 weights; the fourth input channel
//     is a validity mask (1 inside the image, 0 in the padding ring) whose weights carry -sum_c w*mean_c/std_c
//     per tap, which keeps zero padding in NORMALISED space exact; the folded BN bias rides on the centre tap;
//   * ReLU + max-pool run on the accumulator registers (a lane owns one output channel; the 3x3 windows need
//     only lane-local maxima plus 2-4 values from lane^32) and the pooled map is written back as f16 hi/lo planes
//     with a 144-byte pixel pitch (conflict-free ds_read_b128 for 16 consecutive pixels);
//   * layer-1 B fragments (112 VGPRs) stay in registers for the whole layer; layers 2-3 stream theirs from L2.
// The reference transposes H and W before the network (models.py:185-188); all layers are symmetric in the two
// spatial dims, so the kernel works on the frame as rasterised and the host packer swaps the two kernel axes.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <new>
#include <string>
#include <vector>

#include "../../include/srlhip.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4v __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kImg = 64, kCh = 64;
constexpr int kS1 = 14;                 // layer-1 k-steps: K = 7 rows x (8 pixel slots x 4 channels) = 224
constexpr int kS2 = 36;                 // layers 2/3:      K = 9 taps x 64 channels = 576
constexpr float kLoScale = 2048.f, kLoInv = 1.f / 2048.f;
constexpr float kW1Scale = 256.f, kW1Inv = 1.f / 256.f;     // keeps the layer-1 weights in f16's normal range
constexpr float kF16Max = 65504.f;
constexpr float kMean[3] = {0.485f, 0.456f, 0.406f}, kStd[3] = {0.229f, 0.224f, 0.225f};

// ---- LDS map (bytes) -------------------------------------------------------------------------------
constexpr int IN_PITCH = 72 * 8;                 // padded input row: 72 pixels x (R, G, B, mask) f16
constexpr int IN_BYTES = 70 * IN_PITCH;          // 3-pixel zero ring around 64x64
constexpr int PX = 144;                          // pixel pitch of the f16 activation planes (128 B + 16 B skew)
constexpr int A2_PLANE = 257 * PX;               // 16x16 pixels + one all-zero pixel (index 256)
constexpr int A2H = IN_BYTES, A2L = A2H + A2_PLANE;
constexpr int A3_PLANE = 50 * PX;                // 7x7 pixels + one all-zero pixel (index 49)
constexpr int A3H = A2L + A2_PLANE, A3L = A3H + A3_PLANE;
constexpr int XCH = A3L + A3_PLANE;              // [2][8][64] f32: conv-2 row 8 handed to the upper-half waves' neighbours
constexpr int PART = XCH + 4096;                 // [2][8][64] f32: layer-3 partial sums of the second K half
constexpr int FEAT = PART + 4096;                // [64] f32 pooled features
constexpr int LDS_TOTAL = FEAT + 256;
static_assert(IN_BYTES % 16 == 0 && A2_PLANE % 16 == 0 && A3_PLANE % 16 == 0, "LDS planes must stay 16-byte aligned");
static_assert(LDS_TOTAL <= 160 * 1024, "encoder LDS map exceeds one CU");

constexpr size_t kPack1Bytes = 2 * kS1 * 64 * 32, kPack2Bytes = 2 * kS2 * 64 * 32;

struct EncParams {
    const uint8_t *images;      // [n][64][64][3]
    int n;
    const char *b1, *b2, *b3;   // packed B fragments: [n-half][k-step][lane][8 hi | 8 lo] f16
    const float *bias2, *bias3, *fcw, *fcb;
    int state_dim;
    float *out;                 // [n][state_dim]
    int *status;                // bit 0: an activation left f16's range
};

extern __shared__ __attribute__((aligned(16))) char enc_lds[];

__device__ __forceinline__ f32x16 mfma16(half8 a, half8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ half8 lds16(int off) { return *reinterpret_cast<const half8 *>(enc_lds + off); }

// f32 -> (hi, lo) f16 planes at byte offset `off` inside the plane pair starting at hbase / lbase
__device__ __forceinline__ void store_split(int hbase, int lbase, int off, float v, bool &ovf) {
    ovf |= !(v < kF16Max);
    const _Float16 hi = (_Float16)v;
    const _Float16 lo = (_Float16)((v - (float)hi) * kLoScale);
    *reinterpret_cast<_Float16 *>(enc_lds + hbase + off) = hi;
    *reinterpret_cast<_Float16 *>(enc_lds + lbase + off) = lo;
}

// two conv-1 output rows (32 pixels x this wave's 32 channels each), ReLU applied
__device__ __forceinline__ void conv1_pair(const half8 (&Bh)[kS1], const half8 (&Bl)[kS1], int ra, int rb, int lane_base,
                                           f32x16 &v0, f32x16 &v1) {
    f32x16 a0h = {0}, a0l = {0}, a1h = {0}, a1l = {0};
    const int base0 = 2 * ra * IN_PITCH + lane_base, base1 = 2 * rb * IN_PITCH + lane_base;
#pragma unroll
    for (int s = 0; s < kS1; s++) {
        const int off = (s >> 1) * IN_PITCH + (s & 1) * 32;       // kernel row s/2, pixel slots 4(s&1)+2h, +1
        const half8 x0 = lds16(base0 + off), x1 = lds16(base1 + off);
        a0h = mfma16(x0, Bh[s], a0h);
        a0l = mfma16(x0, Bl[s], a0l);
        a1h = mfma16(x1, Bh[s], a1h);
        a1l = mfma16(x1, Bl[s], a1l);
    }
#pragma unroll
    for (int r = 0; r < 16; r++) {
        v0[r] = fmaxf(0.f, (a0h[r] + a0l[r] * kLoInv) * kW1Inv);
        v1[r] = fmaxf(0.f, (a1h[r] + a1l[r] * kLoInv) * kW1Inv);
    }
}

__global__ __launch_bounds__(256, 1) void encoder_fwd_k(EncParams P) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int mh = wave >> 1, nh = wave & 1;      // which half of the pixels / of the output channels this wave owns
    const int j = lane & 31, h = lane >> 5;       // MFMA lane coordinates: column (channel) j, k / row half h
    const int ch = 32 * nh + j;                   // the output channel this lane owns in every layer
    bool ovf = false;

    // -- once per workgroup: zero the padded input (ring stays zero), the all-zero pixels; layer-1 B fragments
    for (int o = tid * 16; o < IN_BYTES; o += 256 * 16) *reinterpret_cast<uint4 *>(enc_lds + o) = make_uint4(0, 0, 0, 0);
    if (tid < 9) {
        *reinterpret_cast<uint4 *>(enc_lds + A2H + 256 * PX + tid * 16) = make_uint4(0, 0, 0, 0);
        *reinterpret_cast<uint4 *>(enc_lds + A2L + 256 * PX + tid * 16) = make_uint4(0, 0, 0, 0);
        *reinterpret_cast<uint4 *>(enc_lds + A3H + 49 * PX + tid * 16) = make_uint4(0, 0, 0, 0);
        *reinterpret_cast<uint4 *>(enc_lds + A3L + 49 * PX + tid * 16) = make_uint4(0, 0, 0, 0);
    }
    const float bias2 = P.bias2[ch], bias3 = P.bias3[ch];
    __syncthreads();

    for (int img = blockIdx.x; img < P.n; img += gridDim.x) {
        // ---- phase 0: uint8 frame -> f16 (R, G, B, 1) pixels inside the zero ring ----------------------------
        {
            const uint4 *src = reinterpret_cast<const uint4 *>(P.images + (size_t)img * (kImg * kImg * 3));
            const uint4 r0 = src[tid], r1 = src[tid + 256], r2 = src[tid + 512];
            uint4 *raw = reinterpret_cast<uint4 *>(enc_lds + A2H);         // the layer-2 planes are free right now
            raw[tid] = r0; raw[tid + 256] = r1; raw[tid + 512] = r2;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int q = tid + 256 * i;                                   // four consecutive pixels = 12 bytes
            const uint32_t *w = reinterpret_cast<const uint32_t *>(enc_lds + A2H + 12 * q);
            const uint32_t w0 = w[0], w1 = w[1], w2 = w[2];
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

        // ---- layer 1: conv7x7/2 + ReLU + maxpool3/2 p1 -> A2 planes (16x16x64 f16 hi/lo) ----------------------
        {
            // this wave's layer-1 B fragments (112 VGPRs) stay in registers for the whole layer
            half8 B1h[kS1], B1l[kS1];
#pragma unroll
            for (int s = 0; s < kS1; s++) {
                const half8 *p = reinterpret_cast<const half8 *>(P.b1 + ((size_t)(nh * kS1 + s) * 64 + lane) * 32);
                B1h[s] = p[0];
                B1l[s] = p[1];
            }
            const int lane_base = j * 16 + h * 16;
            f32x16 carry, v0, v1;
            if (mh == 0) {
#pragma unroll
                for (int r = 0; r < 16; r++) carry[r] = 0.f;               // conv row -1 is pool padding (values are >= 0)
            } else {
                conv1_pair(B1h, B1l, 15, 15, lane_base, v0, carry);
            }
#pragma unroll 1
            for (int p = 0; p < 8; p++) {
                const int prow = 8 * mh + p;                               // pooled row <- conv rows 2p-1, 2p, 2p+1
                conv1_pair(B1h, B1l, 2 * prow, 2 * prow + 1, lane_base, v0, v1);
                f32x16 m;
#pragma unroll
                for (int r = 0; r < 16; r++) m[r] = fmaxf(carry[r], fmaxf(v0[r], v1[r]));
                carry = v1;
                // a lane holds pixels x = 8g + 4h + r (register 4g + r); the pixel left of its 4-group is lane^32's
                const float t3 = __shfl_xor(m[3], 32), t7 = __shfl_xor(m[7], 32), t11 = __shfl_xor(m[11], 32),
                            t15 = __shfl_xor(m[15], 32);
                const float left[4] = {h ? t3 : 0.f, h ? t7 : t3, h ? t11 : t7, h ? t15 : t11};
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const float ka = fmaxf(fmaxf(m[4 * g], m[4 * g + 1]), left[g]);
                    const float kb = fmaxf(fmaxf(m[4 * g + 1], m[4 * g + 2]), m[4 * g + 3]);
                    const int k = 4 * g + 2 * h;
                    store_split(A2H, A2L, (prow * 16 + k) * PX + ch * 2, ka, ovf);
                    store_split(A2H, A2L, (prow * 16 + k + 1) * PX + ch * 2, kb, ovf);
                }
            }
        }
        __syncthreads();

        // ---- layer 2: conv3x3 p1 + ReLU + maxpool3/2 -> A3 planes (7x7x64) -----------------------------------
        {
            f32x16 acc[4][2];
#pragma unroll
            for (int t = 0; t < 4; t++)
#pragma unroll
                for (int r = 0; r < 16; r++) { acc[t][0][r] = 0.f; acc[t][1][r] = 0.f; }
            const int row = j >> 4, ox = j & 15;                           // tile = conv rows 2T, 2T+1 x 16 columns
            const char *bp = P.b2 + ((size_t)(nh * kS2) * 64 + lane) * 32;
            half8 Bh = reinterpret_cast<const half8 *>(bp)[0], Bl = reinterpret_cast<const half8 *>(bp)[1];
#pragma unroll 1
            for (int tap = 0; tap < 9; tap++) {
                const int ky = tap / 3, kx = tap - 3 * ky;
                int addr[4];
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    const int sy = 2 * (4 * mh + t) + row + ky - 1, sx = ox + kx - 1;
                    const bool ok = (unsigned)sy < 16u && (unsigned)sx < 16u;
                    addr[t] = (ok ? sy * 16 + sx : 256) * PX + h * 16;
                }
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int s = 4 * tap + q;
                    half8 Bhn = Bh, Bln = Bl;
                    if (s + 1 < kS2) {
                        const half8 *pn = reinterpret_cast<const half8 *>(bp + (size_t)(s + 1) * 64 * 32);
                        Bhn = pn[0];
                        Bln = pn[1];
                    }
#pragma unroll
                    for (int t = 0; t < 4; t++) {
                        const half8 ah = lds16(A2H + addr[t] + q * 32), al = lds16(A2L + addr[t] + q * 32);
                        acc[t][0] = mfma16(ah, Bh, acc[t][0]);
                        acc[t][1] = mfma16(ah, Bl, acc[t][1]);
                        acc[t][1] = mfma16(al, Bh, acc[t][1]);
                    }
                    Bh = Bhn;
                    Bl = Bln;
                }
            }
            // bias + ReLU; registers 0..7 = conv row 2T, 8..15 = row 2T+1; pixel x = 4h + (r & 3) + 8 ((r >> 2) & 1)
#pragma unroll
            for (int t = 0; t < 4; t++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[t][0][r] = fmaxf(0.f, acc[t][0][r] + acc[t][1][r] * kLoInv + bias2);
            float *xch = reinterpret_cast<float *>(enc_lds + XCH) + nh * 8 * 64 + lane;
            if (mh == 1) {                                                 // conv row 8 completes pooled row 3 of the other half
#pragma unroll
                for (int r = 0; r < 8; r++) xch[r * 64] = acc[0][0][r];
            }
            __syncthreads();
#pragma unroll
            for (int t = 0; t < 4; t++) {
                if (t == 3 && mh == 1) break;                              // pooled row 7 does not exist
                float w[8];
#pragma unroll
                for (int r = 0; r < 8; r++) {
                    const float below = t < 3 ? acc[t < 3 ? t + 1 : 3][0][r] : xch[r * 64];
                    w[r] = fmaxf(fmaxf(acc[t][0][r], acc[t][0][r + 8]), below);
                }
                const float t0 = __shfl_xor(w[0], 32), t4 = __shfl_xor(w[4], 32);
                const int prow = 4 * mh + t;
                const float ka = fmaxf(fmaxf(w[0], w[1]), w[2]);           // x = 4h .. 4h+2      -> k = 2h
                const float kb = fmaxf(fmaxf(w[2], w[3]), h ? t4 : t0);    // x = 4h+2 .. 4h+4    -> k = 2h+1
                const float kc = fmaxf(fmaxf(w[4], w[5]), w[6]);           // x = 8+4h .. 8+4h+2  -> k = 4+2h
                store_split(A3H, A3L, (prow * 7 + 2 * h) * PX + ch * 2, ka, ovf);
                store_split(A3H, A3L, (prow * 7 + 2 * h + 1) * PX + ch * 2, kb, ovf);
                store_split(A3H, A3L, (prow * 7 + 4 + 2 * h) * PX + ch * 2, kc, ovf);
                if (h == 0) {
                    const float kd = fmaxf(fmaxf(w[6], w[7]), t4);         // x = 10, 11, 12      -> k = 5
                    store_split(A3H, A3L, (prow * 7 + 5) * PX + ch * 2, kd, ovf);
                }
            }
        }
        __syncthreads();

        // ---- layer 3: conv3x3/2 p1 (7x7 -> 4x4) + ReLU + maxpool3/2 (-> 1x1); K split over the two wave halves ----
        {
            f32x16 c0 = {0}, c1 = {0};
            const bool pix = j < 16;
            const int oy = (j >> 2) & 3, ox = j & 3;
            const char *bp = P.b3 + ((size_t)(nh * kS2 + 18 * mh) * 64 + lane) * 32;
            half8 Bh = reinterpret_cast<const half8 *>(bp)[0], Bl = reinterpret_cast<const half8 *>(bp)[1];
#pragma unroll
            for (int ii = 0; ii < 18; ii++) {
                const int s = 18 * mh + ii, tap = s >> 2, q = s & 3;
                const int ky = tap / 3, kx = tap - 3 * ky;
                const int sy = 2 * oy + ky - 1, sx = 2 * ox + kx - 1;
                const bool ok = pix && (unsigned)sy < 7u && (unsigned)sx < 7u;
                const int addr = (ok ? sy * 7 + sx : 49) * PX + h * 16 + q * 32;
                half8 Bhn = Bh, Bln = Bl;
                if (ii + 1 < 18) {
                    const half8 *pn = reinterpret_cast<const half8 *>(bp + (size_t)(ii + 1) * 64 * 32);
                    Bhn = pn[0];
                    Bln = pn[1];
                }
                const half8 ah = lds16(A3H + addr), al = lds16(A3L + addr);
                c0 = mfma16(ah, Bh, c0);
                c1 = mfma16(ah, Bl, c1);
                c1 = mfma16(al, Bh, c1);
                Bh = Bhn;
                Bl = Bln;
            }
            float *part = reinterpret_cast<float *>(enc_lds + PART) + nh * 8 * 64 + lane;
            if (mh == 1) {
#pragma unroll
                for (int r = 0; r < 8; r++) part[r * 64] = c0[r] + c1[r] * kLoInv;
            }
            __syncthreads();
            if (mh == 0) {
                // registers 0..3 -> output pixels 4h + r, 4..7 -> 8 + 4h + (r - 4); the 3x3 pool window is
                // pixels {0,1,2,4,5,6,8,9,10}
                float m = 0.f;
#pragma unroll
                for (int r = 0; r < 8; r++) {
                    const float v = fmaxf(0.f, c0[r] + c1[r] * kLoInv + part[r * 64] + bias3);
                    const bool in_window = (r & 3) != 3 && (h == 0 || r < 4);
                    if (in_window) m = fmaxf(m, v);
                }
                m = fmaxf(m, __shfl_xor(m, 32));
                if (h == 0) reinterpret_cast<float *>(enc_lds + FEAT)[ch] = m;
            }
        }
        __syncthreads();

        // ---- FC: state = fc_w . features + fc_b ----------------------------------------------------------------
        {
            const float *feat = reinterpret_cast<const float *>(enc_lds + FEAT);
            for (int sd = tid; sd < P.state_dim; sd += 256) {
                float a = P.fcb[sd];
                const float *wr = P.fcw + (size_t)sd * kCh;
#pragma unroll 8
                for (int c = 0; c < kCh; c++) a = fmaf(wr[c], feat[c], a);
                P.out[(size_t)img * P.state_dim + sd] = a;
            }
        }
        // no barrier needed here: the next writers of FEAT / PART / XCH / A3 sit behind the barriers of the next image
    }
    if (ovf) atomicOr(P.status, 1);
}

// ---------------------------------------------------------------------------------------------- host side
void split_f16(float v, _Float16 &hi, _Float16 &lo) {
    hi = (_Float16)v;
    lo = (_Float16)((v - (float)hi) * kLoScale);
}

// conv1_w [64][3][7][7] (torch OIHW, BatchNorm folded), conv1_b [64].  The network sees the frame with its two
// spatial axes swapped (models.py:185-188), so the tap at frame offset (ky, kx) is torch's w[o][c][kx][ky].
void pack_layer1(const float *w, const float *b, _Float16 *out) {
    for (int nh = 0; nh < 2; nh++)
        for (int s = 0; s < kS1; s++)
            for (int lane = 0; lane < 64; lane++)
                for (int e = 0; e < 8; e++) {
                    const int o = 32 * nh + (lane & 31), k = 16 * s + 8 * (lane >> 5) + e;
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
                    _Float16 *dst = out + ((size_t)(nh * kS1 + s) * 64 + lane) * 16;
                    split_f16((float)(v * (double)kW1Scale), dst[e], dst[8 + e]);
                }
}
// conv_w [64][64][3][3] (torch OIHW, BatchNorm folded): k = (ky * 3 + kx) * 64 + c
void pack_layer3x3(const float *w, _Float16 *out) {
    for (int nh = 0; nh < 2; nh++)
        for (int s = 0; s < kS2; s++)
            for (int lane = 0; lane < 64; lane++)
                for (int e = 0; e < 8; e++) {
                    const int o = 32 * nh + (lane & 31), k = 16 * s + 8 * (lane >> 5) + e;
                    const int tap = k / 64, c = k % 64, ky = tap / 3, kx = tap % 3;
                    _Float16 *dst = out + ((size_t)(nh * kS2 + s) * 64 + lane) * 16;
                    split_f16(w[((o * 64 + c) * 3 + kx) * 3 + ky], dst[e], dst[8 + e]);
                }
}

}  // namespace

struct srlhip_encoder {
    int device_id, state_dim;
    char *d_pack;          // b1 | b2 | b3
    float *d_f32;          // bias2[64] bias3[64] fcb[state_dim] fcw[state_dim][64]
    int *d_status;
    int num_cus;
    std::string err;
    int fail(int code, const std::string &m) { err = m; return code; }
};

namespace {
thread_local std::string g_enc_create_error;

bool supported_shape(int img_h, int img_w, int n_channels) { return img_h == kImg && img_w == kImg && n_channels == 3; }
}  // namespace

extern "C" {

size_t srlhip_encoder_pack_bytes(void) { return kPack1Bytes + 2 * kPack2Bytes; }

int srlhip_encoder_pack(const float *conv1_w, const float *conv1_b, const float *conv2_w, const float *conv3_w,
                        void *out, size_t out_bytes) {
    if (!conv1_w || !conv1_b || !conv2_w || !conv3_w || !out || out_bytes < srlhip_encoder_pack_bytes()) return SRLHIP_EINVAL;
    char *p = static_cast<char *>(out);
    pack_layer1(conv1_w, conv1_b, reinterpret_cast<_Float16 *>(p));
    pack_layer3x3(conv2_w, reinterpret_cast<_Float16 *>(p + kPack1Bytes));
    pack_layer3x3(conv3_w, reinterpret_cast<_Float16 *>(p + kPack1Bytes + kPack2Bytes));
    return SRLHIP_OK;
}

int srlhip_encoder_supported(int32_t img_h, int32_t img_w, int32_t n_channels) { return supported_shape(img_h, img_w, n_channels) ? 1 : 0; }

int srlhip_encoder_create(int32_t device_id, int32_t img_h, int32_t img_w, int32_t n_channels, int32_t state_dim,
                          const float *conv1_w, const float *conv1_b, const float *conv2_w, const float *conv2_b,
                          const float *conv3_w, const float *conv3_b, const float *fc_w, const float *fc_b,
                          srlhip_encoder_handle *out) {
    if (!out) return SRLHIP_EINVAL;
    *out = nullptr;
    if (!supported_shape(img_h, img_w, n_channels) || state_dim < 1 || !conv2_b || !conv3_b || !fc_w || !fc_b) {
        g_enc_create_error = "srlhip_encoder_create: the fused encoder covers 64x64x3 frames (CustomCNN) and state_dim >= 1";
        return SRLHIP_ENOTSUP;
    }
    std::vector<char> pack(srlhip_encoder_pack_bytes());
    if (srlhip_encoder_pack(conv1_w, conv1_b, conv2_w, conv3_w, pack.data(), pack.size()) != SRLHIP_OK) {
        g_enc_create_error = "srlhip_encoder_create: null weight pointer";
        return SRLHIP_EINVAL;
    }
    srlhip_encoder *e = new (std::nothrow) srlhip_encoder();
    if (!e) return SRLHIP_ENOMEM;
    e->device_id = device_id; e->state_dim = state_dim; e->d_pack = nullptr; e->d_f32 = nullptr; e->d_status = nullptr;
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
    const size_t nf = 128 + (size_t)state_dim * (1 + kCh);
    std::vector<float> f(nf);
    memcpy(f.data(), conv2_b, 64 * sizeof(float));
    memcpy(f.data() + 64, conv3_b, 64 * sizeof(float));
    memcpy(f.data() + 128, fc_b, state_dim * sizeof(float));
    memcpy(f.data() + 128 + state_dim, fc_w, (size_t)state_dim * kCh * sizeof(float));
    ENC_CHECK(hipMalloc(reinterpret_cast<void **>(&e->d_pack), pack.size()));
    ENC_CHECK(hipMalloc(reinterpret_cast<void **>(&e->d_f32), nf * sizeof(float)));
    ENC_CHECK(hipMalloc(reinterpret_cast<void **>(&e->d_status), sizeof(int)));
    ENC_CHECK(hipMemcpy(e->d_pack, pack.data(), pack.size(), hipMemcpyHostToDevice));
    ENC_CHECK(hipMemcpy(e->d_f32, f.data(), nf * sizeof(float), hipMemcpyHostToDevice));
    ENC_CHECK(hipMemset(e->d_status, 0, sizeof(int)));
    ENC_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(encoder_fwd_k), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL));
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
    EncParams p;
    p.images = images_dev; p.n = n;
    p.b1 = e->d_pack; p.b2 = e->d_pack + kPack1Bytes; p.b3 = e->d_pack + kPack1Bytes + kPack2Bytes;
    p.bias2 = e->d_f32; p.bias3 = e->d_f32 + 64; p.fcb = e->d_f32 + 128; p.fcw = e->d_f32 + 128 + e->state_dim;
    p.state_dim = e->state_dim; p.out = states_dev; p.status = e->d_status;
    const int grid = n < e->num_cus ? n : e->num_cus;
    hipLaunchKernelGGL(encoder_fwd_k, dim3(grid), dim3(256), LDS_TOTAL, static_cast<hipStream_t>(hip_stream), p);
    rc = hipGetLastError();
    if (rc != hipSuccess) return e->fail(SRLHIP_EHIP, std::string("encoder_fwd_k launch: ") + hipGetErrorString(rc));
    return SRLHIP_OK;
}

int srlhip_encoder_overflow(srlhip_encoder_handle e, int32_t *flag) {
    if (!e || !flag) return SRLHIP_EINVAL;
    hipError_t rc = hipSetDevice(e->device_id);
    if (rc == hipSuccess) rc = hipDeviceSynchronize();
    int v = 0;
    if (rc == hipSuccess) rc = hipMemcpy(&v, e->d_status, sizeof(int), hipMemcpyDeviceToHost);
    if (rc != hipSuccess) return e->fail(SRLHIP_EHIP, std::string("srlhip_encoder_overflow: ") + hipGetErrorString(rc));
    *flag = v & 1;
    return SRLHIP_OK;
}

int srlhip_encoder_destroy(srlhip_encoder_handle e) {
    if (!e) return SRLHIP_OK;
    (void)hipSetDevice(e->device_id);
    if (e->d_pack) (void)hipFree(e->d_pack);
    if (e->d_f32) (void)hipFree(e->d_f32);
    if (e->d_status) (void)hipFree(e->d_status);
    delete e;
    return SRLHIP_OK;
}

const char *srlhip_encoder_last_error(srlhip_encoder_handle e) { return e ? e->err.c_str() : g_enc_create_error.c_str(); }

}  // extern "C"
