// pgs_row_timing.hip — what one projected-Gauss-Seidel row of the Kuka kernel costs as an instruction sequence on gfx950.
//   hipcc -O3 --offload-arch=gfx950 profiles/probes/pgs_row_timing.hip -o build/pgs_row_timing && build/pgs_row_timing
// One wavefront per SIMD (256 threads on one CU), 12 rows x 150 sweeps x REP steps of each variant, s_memrealtime ticks (100 MHz).
// The sequences are the ones of csrc/kuka_tree.hpp (sweeps_free / sweeps_contacts): t = clamp01(cs + acc); the own accumulator
// restarts; acc += n_j * bcast_j(t) [; accB += m_j * bcast_j(t)].  A DPP operand written by the previous VALU instruction needs two
// wait states: V0 fills them with the restart FMA and an s_nop, V1 with the restart FMA and the PREVIOUS row's second fmac.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define ROW_V0(J) "v_add_f64 %[t], %[cs], %[acc] clamp\n\tv_fma_f64 %[acc], -%[e], %[acc], %[acc]\n\ts_nop 0\n\t" \
                  "v_fmac_f64_dpp %[acc], %[t], %[n] row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n\t"
#define ROW_V2(J) "v_add_f64 %[t], %[cs], %[acc] clamp\n\tv_fma_f64 %[acc], -%[e], %[acc], %[acc]\n\ts_nop 0\n\t" \
                  "v_fmac_f64_dpp %[acc], %[t], %[n] row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n\t" \
                  "v_fmac_f64_dpp %[accb], %[t], %[m] row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n\ts_nop 0\n\t"
// pipelined: rows alternate between two t registers; the previous row's accB update sits in the hazard slot
#define ROW_V1(J, T, TP, JP) "v_add_f64 %[" #T "], %[cs], %[acc] clamp\n\tv_fma_f64 %[acc], -%[e], %[acc], %[acc]\n\t" \
                  "v_fmac_f64_dpp %[accb], %[" #TP "], %[m] row_newbcast:" #JP " row_mask:0xf bank_mask:0xf\n\t" \
                  "v_fmac_f64_dpp %[acc], %[" #T "], %[n] row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n\t"
#define ROW_V3(J) "v_add_f64 %[t], %[cs], %[acc] clamp\n\ts_nop 1\n\t" \
                  "v_fmac_f64_dpp %[acc], %[t], %[n] row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n\t"
#define ROW_V4(J) "v_add_f64 %[t], %[cs], %[acc] clamp\n\tv_fma_f64 %[acc], -%[e], %[acc], %[acc]\n\tv_mov_b32 %[f], %[f]\n\t" \
                  "v_fmac_f64_dpp %[acc], %[t], %[n] row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n\t"
#define ALL12(R) R(0) R(1) R(2) R(3) R(4) R(5) R(6) R(7) R(8) R(9) R(10) R(11)
// the free sweep as it is: the button's rows 12..14 ride on rows 0..2 — a second fmac IN the accumulator's chain
#define ROW_RIDE(J, JB) "v_add_f64 %[t], %[cs], %[acc] clamp\n\tv_fma_f64 %[acc], -%[e], %[acc], %[acc]\n\ts_nop 0\n\t" \
                  "v_fmac_f64_dpp %[acc], %[t], %[n] row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n\t" \
                  "v_fmac_f64_dpp %[acc], %[t], %[m] row_newbcast:" #JB " row_mask:0xf bank_mask:0xf\n\t"
// the button's three rows as their own chain (accumulator accb, value tb), one of its instructions in each arm row's wait states
#define ROW_SEP(J, FILL) "v_add_f64 %[t], %[cs], %[acc] clamp\n\tv_fma_f64 %[acc], -%[e], %[acc], %[acc]\n\t" FILL \
                  "v_fmac_f64_dpp %[acc], %[t], %[n] row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n\t"
#define B_ADD "v_add_f64 %[tb], %[cs], %[accb] clamp\n\t"
#define B_RST "v_fma_f64 %[accb], -%[e], %[accb], %[accb]\n\t"
#define B_MAC(JB) "v_fmac_f64_dpp %[accb], %[tb], %[m] row_newbcast:" #JB " row_mask:0xf bank_mask:0xf\n\t"

template <int V>
__global__ void rows_k(double *out, long long *ticks, int sweeps, double cs, double e, double n, double m) {
    double acc = threadIdx.x * 1e-3, accb = 0.0, t = 0.0, t2 = 0.0;
    int f = 0;
    __builtin_amdgcn_s_barrier();
    const long long t0 = wall_clock64();
    for (int s = 0; s < sweeps; s++) {
        if constexpr (V == 0) asm volatile(ALL12(ROW_V0) : [acc] "+v"(acc), [t] "=&v"(t) : [cs] "v"(cs), [e] "v"(e), [n] "v"(n));
        if constexpr (V == 2) asm volatile(ALL12(ROW_V2) : [acc] "+v"(acc), [accb] "+v"(accb), [t] "=&v"(t) : [cs] "v"(cs), [e] "v"(e), [n] "v"(n), [m] "v"(m));
        if constexpr (V == 1)
            asm volatile(ROW_V1(0, t, t2, 11) ROW_V1(1, t2, t, 0) ROW_V1(2, t, t2, 1) ROW_V1(3, t2, t, 2) ROW_V1(4, t, t2, 3) ROW_V1(5, t2, t, 4)
                         ROW_V1(6, t, t2, 5) ROW_V1(7, t2, t, 6) ROW_V1(8, t, t2, 7) ROW_V1(9, t2, t, 8) ROW_V1(10, t, t2, 9) ROW_V1(11, t2, t, 10)
                         : [acc] "+v"(acc), [accb] "+v"(accb), [t] "+v"(t), [t2] "+v"(t2) : [cs] "v"(cs), [e] "v"(e), [n] "v"(n), [m] "v"(m));
        if constexpr (V == 3) asm volatile(ALL12(ROW_V3) : [acc] "+v"(acc), [t] "=&v"(t) : [cs] "v"(cs), [n] "v"(n));
        if constexpr (V == 5)
            asm volatile(ROW_RIDE(0, 12) ROW_RIDE(1, 13) ROW_RIDE(2, 14) ROW_V0(3) ROW_V0(4) ROW_V0(5) ROW_V0(6) ROW_V0(7) ROW_V0(8) ROW_V0(9) ROW_V0(10) ROW_V0(11)
                         : [acc] "+v"(acc), [t] "=&v"(t) : [cs] "v"(cs), [e] "v"(e), [n] "v"(n), [m] "v"(m));
        if constexpr (V == 6)
            asm volatile(ROW_SEP(0, B_ADD) ROW_SEP(1, B_RST) ROW_SEP(2, "s_nop 0\n\t") ROW_SEP(3, B_MAC(12)) ROW_SEP(4, B_ADD) ROW_SEP(5, B_RST) ROW_SEP(6, "s_nop 0\n\t")
                         ROW_SEP(7, B_MAC(13)) ROW_SEP(8, B_ADD) ROW_SEP(9, B_RST) ROW_SEP(10, "s_nop 0\n\t") ROW_SEP(11, B_MAC(14))
                         : [acc] "+v"(acc), [accb] "+v"(accb), [t] "=&v"(t), [tb] "+v"(t2) : [cs] "v"(cs), [e] "v"(e), [n] "v"(n), [m] "v"(m));
        if constexpr (V == 4) asm volatile(ALL12(ROW_V4) : [acc] "+v"(acc), [t] "=&v"(t), [f] "+v"(f) : [cs] "v"(cs), [e] "v"(e), [n] "v"(n));
    }
    const long long t1 = wall_clock64();
    out[threadIdx.x] = acc + accb + t + t2 + f;
    if (threadIdx.x % 64 == 0) ticks[threadIdx.x / 64] = t1 - t0;
}

template <int V> static void run(const char *what, int threads) {
    double *out; long long *ticks;
    (void)hipMalloc(&out, sizeof(double) * 512); (void)hipMalloc(&ticks, sizeof(long long) * 16);
    const int sweeps = 150 * 200;
    rows_k<V><<<1, threads>>>(out, ticks, sweeps, 0.25, 0.0, 1e-3, 1e-3);
    (void)hipDeviceSynchronize();
    rows_k<V><<<1, threads>>>(out, ticks, sweeps, 0.25, 0.0, 1e-3, 1e-3);
    (void)hipDeviceSynchronize();
    std::vector<long long> t(16); (void)hipMemcpy(t.data(), ticks, sizeof(long long) * 16, hipMemcpyDeviceToHost);
    long long tmax = 0; for (int w = 0; w < threads / 64; w++) tmax = t[w] > tmax ? t[w] : tmax;
    printf("%-78s %d wave(s)/SIMD: %.2f ns per row\n", what, threads <= 256 ? 1 : threads / 256, tmax * 10.0 / ((double)sweeps * 12));
    (void)hipFree(out); (void)hipFree(ticks);
}

int main() {
    for (int threads : {256, 512}) {
        run<0>("V0 free row now: add, restart fma, s_nop 0, fmac_dpp", threads);
        run<4>("V4 the same with a VALU filler (v_mov_b32) instead of the s_nop", threads);
        run<3>("V3 no restart: add, s_nop 1, fmac_dpp (lower bound of the chain)", threads);
        run<5>("V5 free SWEEP now: rows 0..2 carry the button's fmac in the chain (per row of 12)", threads);
        run<6>("V6 free sweep with the button's three rows as a separate interleaved chain (per row of 12)", threads);
        run<2>("V2 contact row now: add, restart fma, s_nop 0, fmac_dpp, fmac_dpp (accB), s_nop 0", threads);
        run<1>("V1 contact row pipelined: add, restart fma, fmac_dpp (accB of the previous row), fmac_dpp", threads);
    }
    return 0;
}
