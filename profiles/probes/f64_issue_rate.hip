// f64_issue_rate.hip — how fast does one gfx950 SIMD issue v_fma_f64, and what is its dependent-issue latency?
//   hipcc -O3 --offload-arch=gfx950 profiles/probes/f64_issue_rate.hip -o build/f64_issue_rate && build/f64_issue_rate
// One workgroup on one CU; W wavefronts (64 = 1 wave on one SIMD, 256 = one per SIMD, 512 = two per SIMD); every wavefront runs
// CH independent chains of `iters` dependent FMAs.  cycles = s_memtime ticks (100 MHz constant clock on gfx950: reported in ns too)
// -> per-FMA cost per SIMD.  The projected Gauss-Seidel sweep of the Kuka kernel (csrc/kuka_tree.hpp, sweeps_free) is one
// dependent chain of 2 f64 ops per row: whether a second wavefront per SIMD can hide anything depends on these two numbers.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int CH, bool F32>
__global__ void chain_k(double *out, long long *ticks, int iters, double a, double b) {
    using T = typename std::conditional<F32, float, double>::type;
    T x[CH];
    for (int c = 0; c < CH; c++) x[c] = (T)(threadIdx.x * 1e-3 + c);
    const T ta = (T)a, tb = (T)b;
    __builtin_amdgcn_s_barrier();
    const long long t0 = wall_clock64();
    for (int i = 0; i < iters; i += 16) {                          // 16 dependent steps per trip: the branch is amortised
#pragma unroll
        for (int k = 0; k < 16; k++)
#pragma unroll
            for (int c = 0; c < CH; c++) {
                if constexpr (F32) x[c] = __builtin_fmaf(x[c], ta, tb);
                else x[c] = __builtin_fma(x[c], ta, tb);
            }
    }
    const long long t1 = wall_clock64();
    T s = 0;
    for (int c = 0; c < CH; c++) s += x[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (double)s;
    if (threadIdx.x % 64 == 0) ticks[threadIdx.x / 64] = t1 - t0;
}

template <int CH, bool F32>
static void run(int threads, int iters) {
    double *out; long long *ticks;
    (void)hipMalloc(&out, sizeof(double) * threads); (void)hipMalloc(&ticks, sizeof(long long) * 16);
    chain_k<CH, F32><<<1, threads>>>(out, ticks, iters, 0.999999, 1e-7);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    chain_k<CH, F32><<<1, threads>>>(out, ticks, iters, 0.999999, 1e-7);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> t(16); (void)hipMemcpy(t.data(), ticks, sizeof(long long) * 16, hipMemcpyDeviceToHost);
    long long tmax = 0; for (int w = 0; w < threads / 64; w++) tmax = t[w] > tmax ? t[w] : tmax;
    const double ns = tmax * 10.0;                                 // 100 MHz
    const int waves_per_simd = threads <= 256 ? 1 : threads / 256;
    const double fma_per_simd = (double)iters * CH * waves_per_simd;
    printf("%s chains=%d waves=%d (%d/SIMD): %.2f ns per FMA per SIMD (%.2f ns per dependent step), event %.3f ms\n", F32 ? "f32" : "f64", CH,
           threads / 64, waves_per_simd, ns / fma_per_simd, ns / iters, ms);
    (void)hipFree(out); (void)hipFree(ticks);
}

int main() {
    const int iters = 1600000;
    for (int threads : {64, 256, 512}) {
        run<1, false>(threads, iters); run<2, false>(threads, iters); run<4, false>(threads, iters); run<8, false>(threads, iters);
    }
    for (int threads : {64, 512}) { run<1, true>(threads, iters); run<8, true>(threads, iters); }
    return 0;
}
