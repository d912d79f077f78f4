// encoder_general.hpp — the layered split-f16 MFMA encoder for every frame shape the fused 64x64x3 kernel
// (encoder.hip) does not cover: 224x224x3 (the reference's RENDER_HEIGHT/WIDTH), 6-channel multi_view frames, ...
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <string>

namespace srlenc {

struct Geometry {
    int H, W, C;                 // frame rows, columns, channels (3 or 6)
    int Hc[3], Wc[3];            // convolution output of layer 1..3
    int Hp[3], Wp[3];            // after that layer's 3x3/2 max-pool
    bool ok;
};
// conv7x7/2 p3 + pool3/2 p1 -> conv3x3 p1 + pool3/2 -> conv3x3/2 p1 + pool3/2 (floor mode, like torch)
Geometry geometry(int img_h, int img_w, int n_channels);

struct General;                  // device buffers + geometry of one encoder handle

// weights in torch layout with the BatchNorms folded (see srlhip.h); fc_w is [state_dim][64 * Wp3 * Hp3] in torch's
// flatten order of the TRANSPOSED frame (channel, frame column, frame row)
int general_create(int device_id, const Geometry &g, int state_dim, const float *conv1_w, const float *conv1_b, const float *conv2_w,
                   const float *conv2_b, const float *conv3_w, const float *conv3_b, const float *fc_w, const float *fc_b,
                   General **out, std::string &err);
int general_forward(General *g, const uint8_t *images_dev, int n, float *states_dev, hipStream_t stream, std::string &err);
int *general_status(General *g);
void general_destroy(General *g);

// layer 1 of the layered path, host-only (srlhip_encoder_pack_first_layer); returns the pre-scale, 0 on bad arguments
float pack_layer1_general(const float *w, const float *b, int n_channels, void *out, size_t out_bytes);

// int8 layer 1 (3-channel frames; csrc/encoder.hip has the description): three balanced base-256 digits of per-channel 24-bit fixed-point
// weights, [channel half][kernel row][digit][lane][16 i8]; zero_slot = the pixel slot of a kernel row that carries no tap (0: fused
// kernel, 7: layered kernels); inv256[o] = 256 / scale_o
constexpr int kI8Steps = 7, kI8Digits = 3, kMaskI8 = 127, kWeightTopI8 = 127 * 65536 + 127 * 256 + 127;
constexpr size_t kPackI8Bytes = 2 * kI8Steps * kI8Digits * 64 * 16;
void pack_layer1_i8(const float *w, const float *b, int zero_slot, int8_t *out, float *inv256);

// shared with encoder.hip's packers
void split_f16(float v, _Float16 &hi, _Float16 &lo);
float pick_scale(double wmax);
float pack_layer3x3(const float *w, _Float16 *out);
constexpr size_t kPack3x3Bytes = 2 * 36 * 64 * 32;

}  // namespace srlenc
