// Micro-probe (run on the GPU box): what paces one wave's v_mfma_f32_32x32x16_f16 stream in the encoder's layer-2 loop?
// Variants of the same 36 x 12 MFMA loop: bare MFMAs / + LDS A-fragment reads / + streamed B fragments from global.
// Build: hipcc -O3 --offload-arch=gfx950 mfma_probe.hip -o mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
extern __shared__ __attribute__((aligned(16))) char lds[];

template <int MODE, int CHAINS>
__global__ __launch_bounds__(256, 1) void probe(const char *bglob, float *out, long long *cyc) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int o = threadIdx.x * 16; o < 80000; o += 256 * 16) *reinterpret_cast<uint4 *>(lds + o) = make_uint4(0x3c003c00, 0x3c003c00, 0x3c003c00, 0x3c003c00);
    __syncthreads();
    f32x16 acc[CHAINS];
    for (int t = 0; t < CHAINS; t++) for (int r = 0; r < 16; r++) acc[t][r] = 0.f;
    half8 b0, b1;
    for (int e = 0; e < 8; e++) { b0[e] = (_Float16)(0.001f * lane + e); b1[e] = (_Float16)(0.002f * lane - e); }
    half8 a[CHAINS], al[CHAINS];
    for (int t = 0; t < CHAINS; t++) { a[t] = b0; al[t] = b1; }
    const char *bp = bglob + ((size_t)(wave & 1) * 36 * 64 + lane) * 32;
    half8 Rh[6], Rl[6];
    if (MODE >= 2) for (int s = 0; s < 6; s++) { const half8 *p = (const half8 *)(bp + (size_t)s * 2048); Rh[s] = p[0]; Rl[s] = p[1]; }
    long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int ky = 0; ky < 3; ky++) {
#pragma unroll
        for (int i = 0; i < 12; i++) {
            half8 na[CHAINS], nl[CHAINS];
            if (MODE >= 1) {
#pragma unroll
                for (int t = 0; t < CHAINS; t++) {
                    const int addr = ((ky * 12 + i + t * 16 + (lane & 31)) % 256) * 144 + (lane >> 5) * 16;
                    na[t] = *(const half8 *)(lds + addr); nl[t] = *(const half8 *)(lds + 37008 + addr);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            const half8 Bh = MODE >= 2 ? Rh[i % 6] : b0, Bl = MODE >= 2 ? Rl[i % 6] : b1;
#pragma unroll
            for (int t = 0; t < CHAINS; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[t], Bh, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < CHAINS; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[t], Bl, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < CHAINS; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[t], Bh, acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (MODE >= 2) {
                const int sn = 12 * ky + i + 6;
                const half8 *p = (const half8 *)(bp + (size_t)(sn < 36 ? sn : 35) * 2048);
                Rh[i % 6] = p[0]; Rl[i % 6] = p[1];
            }
            if (MODE >= 1) {
#pragma unroll
                for (int t = 0; t < CHAINS; t++) { a[t] = na[t]; al[t] = nl[t]; }
            }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int t = 0; t < CHAINS; t++) for (int r = 0; r < 16; r++) s += acc[t][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (lane == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}

template <int MODE, int CHAINS>
void run(const char *name, const char *b, float *out, long long *cyc, int grid) {
    hipFuncSetAttribute((const void *)probe<MODE, CHAINS>, hipFuncAttributeMaxDynamicSharedMemorySize, 81920);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<MODE, CHAINS><<<grid, 256, 81920>>>(b, out, cyc);
    hipEventRecord(e0);
    for (int i = 0; i < 20; i++) probe<MODE, CHAINS><<<grid, 256, 81920>>>(b, out, cyc);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[4]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    const int nm = 36 * 3 * CHAINS;
    printf("%-34s grid %4d chains %d: %6lld ticks/wave for %d MFMAs = %.1f ticks/MFMA; launch %.1f us\n", name, grid, CHAINS, h[0], nm,
           (double)h[0] / nm, ms * 1000 / 20);
}

int main() {
    char *b; float *out; long long *cyc;
    hipMalloc(&b, 2 * 36 * 64 * 32); hipMemset(b, 0, 2 * 36 * 64 * 32);
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 64);
    for (int grid : {1, 256}) {
        run<0, 4>("bare MFMA, 4 chains", b, out, cyc, grid);
        run<0, 2>("bare MFMA, 2 chains", b, out, cyc, grid);
        run<0, 1>("bare MFMA, 1 chain", b, out, cyc, grid);
        run<1, 4>("+ LDS A fragments", b, out, cyc, grid);
        run<2, 4>("+ LDS A + global B ring", b, out, cyc, grid);
    }
    return 0;
}
