// Issue cost of the instruction sequences the lane-group Kuka solver is made of, one wavefront per SIMD (1024 workgroups of
// 64 lanes), timed with HIP events: ns per instruction of a long unrolled run, dependent chains.
//   hipcc --offload-arch=gfx950 -O3 dpp_probe.hip -o dpp_probe && ./dpp_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int KIND>
__global__ void __launch_bounds__(64) probe_k(double *out, int iters) {
    double acc = threadIdx.x * 1e-3, t = 0.5, n = 1e-9 * (threadIdx.x + 1), cs = 0.25, ep = 0.0, b = 0.0;
    for (int i = 0; i < iters; i++) {
        if (KIND == 0) { REP64(asm volatile("v_fma_f64 %0, %1, %0, %2" : "+v"(acc) : "v"(n), "v"(cs));) }
        if (KIND == 1) { REP64(asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(t), "v"(n));) }
        if (KIND == 2) { REP64(asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(n));) }
        if (KIND == 3) { REP64(asm volatile("v_mov_b64_dpp %1, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fma_f64 %0, %1, %2, %0" : "+v"(acc), "=&v"(b) : "v"(n));) }
        if (KIND == 4) { REP64(asm volatile("v_add_f64 %1, %2, %0 clamp\n\tv_fma_f64 %0, -%4, %0, %0\n\ts_nop 0\n\tv_fmac_f64_dpp %0, %1, %3 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(acc), "=&v"(t) : "v"(cs), "v"(n), "v"(ep));) }
        if (KIND == 5) { REP64(asm volatile("v_add_f64 %1, %2, %0 clamp\n\tv_fma_f64 %0, -%4, %0, %0\n\tv_fma_f64 %0, %1, %3, %0" : "+v"(acc), "=&v"(t) : "v"(cs), "v"(n), "v"(ep));) }
        if (KIND == 6) { int ia = threadIdx.x, ib = 0; REP64(asm volatile("v_mov_b32_dpp %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_u32_e32 %0, %1, %0" : "+v"(ia), "=&v"(ib));) acc += ia; }
        if (KIND == 9) { double a1 = acc + 1, a2 = acc + 2, a3 = acc + 3; REP64(asm volatile("v_fma_f64 %0, %4, %0, %5\n\tv_fma_f64 %1, %4, %1, %5\n\tv_fma_f64 %2, %4, %2, %5\n\tv_fma_f64 %3, %4, %3, %5" : "+v"(acc), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(n), "v"(cs));) acc += a1 + a2 + a3; }
        if (KIND == 10) { float f0 = threadIdx.x, f1 = 1.f; REP64(asm volatile("v_fma_f32 %0, %1, %0, %1" : "+v"(f0) : "v"(f1));) acc += f0; }
        if (KIND == 7) { REP64(asm volatile("v_add_f64 %1, %2, %0 clamp\n\tv_fma_f64 %0, -%4, %0, %0\n\ts_nop 0\n\tv_mov_b64_dpp %1, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fma_f64 %0, %1, %3, %0" : "+v"(acc), "=&v"(t) : "v"(cs), "v"(n), "v"(ep));) }
    }
    out[blockIdx.x * 64 + threadIdx.x] = acc + t + b;
}

template <int KIND>
void run(const char *name, int instr_per_rep, double *d, int blocks = 1024) {
    const int iters = 2000;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(probe_k<KIND>, dim3(blocks), dim3(64), 0, 0, d, 10);
    hipDeviceSynchronize();
    hipEventRecord(a, 0);
    hipLaunchKernelGGL(probe_k<KIND>, dim3(blocks), dim3(64), 0, 0, d, iters);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double per_rep = ms * 1e6 / ((double)iters * 64);
    printf("[%4d waves] %-58s %7.2f ns per repetition (%d instr)  = %5.2f ns/instr = %5.1f cycles/instr @2.4GHz\n", blocks, name, per_rep, instr_per_rep,
           per_rep / instr_per_rep, per_rep / instr_per_rep * 2.4);
}

int main() {
    double *d;
    hipMalloc(&d, 1024 * 64 * sizeof(double));
    for (int blocks : {64, 256, 512, 1024, 2048}) {
        run<0>("v_fma_f64 dependent chain", 1, d, blocks);
        run<9>("4 independent v_fma_f64 chains", 4, d, blocks);
        run<10>("v_fma_f32 dependent chain", 1, d, blocks);
        run<4>("PGS row: add-clamp, fma, s_nop 0, fmac_dpp", 3, d, blocks);
    }
    run<0>("v_fma_f64 dependent chain", 1, d);
    run<1>("v_fmac_f64_dpp row_newbcast (acc chain, x fixed)", 1, d);
    run<2>("s_nop 1 + v_fmac_f64_dpp reading the acc it writes", 1, d);
    run<3>("v_mov_b64_dpp + v_fma_f64", 2, d);
    run<4>("PGS row: add-clamp, fma, s_nop 0, fmac_dpp", 3, d);
    run<5>("same row with a plain fma instead of the dpp fmac", 3, d);
    run<6>("v_mov_b32_dpp row_shr + v_add_u32", 2, d);
    run<7>("PGS row with v_mov_b64_dpp + fma instead of fmac_dpp", 4, d);
    return 0;
}
