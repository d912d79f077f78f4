// kuka_group_cm.hip — lane-group Kuka kernels for a runtime model table (srlhip_set_kuka_model): rollout, reset, the settle
// kernel, and the device self-test of the lane-group primitives.
#include "kuka_group_kernels.hpp"

namespace srl {
using namespace kuka;

namespace {
// srlhip_reset with a runtime model table installed: KukaButtonGymEnv.reset by lane groups (the start-state table of
// kuka_reset_k belongs to the baked model).
template <int MODE, bool JOINTS>
__global__ void __launch_bounds__(kGroupBlock)
kuka_group_reset_k(KukaParams p, KukaState s, RngState rs, EpisodeStats st, const uint8_t *mask, const double *host_rand, int rand_stride,
                   float *obs) {
    using namespace grp;
    __shared__ double scratch_all[kGroupEnvs][kScratchDoubles];
    const int64_t n = p.n;
    const int e_raw = blockIdx.x * kGroupEnvs + (int)(threadIdx.x / GL);
    const bool valid = e_raw < p.n && !(mask && !mask[e_raw < p.n ? e_raw : 0]);
    const int e = e_raw < p.n ? e_raw : p.n - 1;
    Lane L; lane_init<true>(L, s.model);
    const bool lead = L.l == 0 && valid;
    using Rng = std::conditional_t<MODE == SRLHIP_RNG_PHILOX, GroupPhilox, typename KRng<MODE>::type>;
    Rng rng0;
    if constexpr (MODE == SRLHIP_RNG_PHILOX) rng0.init(rs.key[e], rs.key[n + e], rs.ctr[e]);
    else krng_load<MODE>(rng0, rs, e, p.n, host_rand ? host_rand + (int64_t)e * rand_stride : nullptr);
    Lane0Rng<Rng> rng_l0{&rng0, lead};
    Env v = {};
    GState g;
    double *objs = valid ? s.objs + e : nullptr;
    if constexpr (MODE == SRLHIP_RNG_MT19937) genv_reset<JOINTS, true>(v, g, L, p.cfg, scratch_all[threadIdx.x / GL], rng_l0, s.starts, s.settled, objs, n);
    else genv_reset<JOINTS, true>(v, g, L, p.cfg, scratch_all[threadIdx.x / GL], rng0, s.starts, s.settled, objs, n);
    if (valid && L.arm) {
        s.d[(D_Q + L.l) * n + e] = g.q; s.d[(D_QD + L.l) * n + e] = g.qd; s.d[(D_SQ + L.l) * n + e] = g.sq; s.d[(D_CQ + L.l) * n + e] = g.cq;
    }
    if (lead) {
#pragma unroll
        for (int k = 0; k < 3; k++) { s.d[(D_EE + k) * n + e] = v.ee[k]; s.d[(D_BPOS + k) * n + e] = v.bpos[k]; s.d[(D_GRIP + k) * n + e] = v.grip[k]; }
        s.d[D_BQ * n + e] = v.bq; s.d[D_BQD * n + e] = v.bqd; s.d[D_BX * n + e] = v.bx; s.d[D_BY * n + e] = v.by;
        s.d[D_BZ * n + e] = v.bz; s.d[D_BSPEED * n + e] = v.bspeed;
        s.i[I_MOTOR * n + e] = v.motor_on; s.i[I_CB * n + e] = v.contact_button; s.i[I_CT * n + e] = v.contact_table;
        s.i[I_COUNTER * n + e] = v.counter; s.i[I_NCONTACT * n + e] = v.n_contacts; s.i[I_NOUT * n + e] = v.n_outside;
        s.i[I_TERM * n + e] = v.terminated;
        if constexpr (MODE == SRLHIP_RNG_PHILOX) rs.ctr[e] = rng0.p.ctr;
        else krng_store<MODE>(rng0, rs, e);
        st.ep_return[e] = 0.0; st.ep_length[e] = 0;
        if (obs) {
            const int od = p.cfg.obs_mode == 1 ? 14 : p.cfg.obs_mode == 2 ? 17 : 3;
            observe(v, p.cfg, obs + (int64_t)e * od, 1);
        }
    }
}

// Settled state of a runtime model table: 500 zero-action steps (kuka_button_gym_env.py:242-247) by the lane-group stepper
// (every group of the wavefront integrates the same env; group 0 publishes, pack_start() layout).
__global__ void __launch_bounds__(kGroupBlock) kuka_group_settle_k(KukaParams p, KukaState s) {
    using namespace grp;
    __shared__ double scratch_all[kGroupEnvs][kScratchDoubles];
    Lane L; lane_init<true>(L, s.model);
    Env e = {};
    GState g;
    g.q = L.arm ? L.q0 : 0.0; g.qd = 0.0; g.sq = 0.0; g.cq = 1.0;
#pragma unroll
    for (int k = 0; k < 3; k++) { e.ee[k] = kEeInit[k]; e.bpos[k] = 0.0; }
    e.bx = kButtonX; e.by = kButtonY; e.bz = L.base_z;
    grefresh<true>(L, g, e);
    const double zero[3] = {0, 0, 0};
    for (int i = 0; i < kNSettleSteps; i++) gphysics_step<true>(e, g, L, p.cfg, scratch_all[threadIdx.x / GL], zero, p.cfg.action_joints != 0, L.q0);
    if (threadIdx.x < GL) {
        double *o = s.settled;
        if (L.arm) { o[L.l] = g.q; o[7 + L.l] = g.qd; o[14 + L.l] = g.sq; o[21 + L.l] = g.cq; }
        if (L.l == 0) {
#pragma unroll
            for (int k = 0; k < 3; k++) { o[28 + k] = e.ee[k]; o[33 + k] = e.grip[k]; }
            o[31] = e.bq; o[32] = e.bqd;
        }
    }
}

// Self-test of the lane-group primitives on the device (tests/test_gpu_group_primitives.py compares with what the host
// emulation of the same source defines): one wavefront, out[k][lane].
constexpr int kProbeRows = 40;
__global__ void __launch_bounds__(64) kuka_group_probe_k(const double *q7, double *out) {
    using namespace grp;
    const int t = threadIdx.x;
    Lane L; lane_init<false>(L, nullptr);
    const double x = 1.5 * t + 0.25;
    int k = 0;
#define SRL_OUT(v) out[(k++) * 64 + t] = (v);
    SRL_OUT((double)L.l)
    SRL_OUT(bcast<3>(x)) SRL_OUT(bcast<15>(x))
    SRL_OUT(shr<1>(x, -1.0)) SRL_OUT(shr<2>(x, -2.0)) SRL_OUT(shr<4>(x, -4.0))
    SRL_OUT((double)ballot(t % 3 == 0)) SRL_OUT(gany(t == 37) ? 1.0 : 0.0) SRL_OUT(wany(t == 37) ? 1.0 : 0.0)
    { double acc = (double)t; fmac_bcast<5>(acc, x, 2.0); SRL_OUT(acc) }
    { double acc = 0.125 * t; const double tt = pgs_row<2>(acc, 0.125 * (t & 15) - 0.25, 0.5, t % 16 == 4 ? 1.0 : 0.0); SRL_OUT(acc) SRL_OUT(tt) }
    { double acc = 0.125 * t; const double tt = pgs_row2<1, 9>(acc, 0.25 * (t & 15) - 0.5, 0.5, t % 16 >= 8 ? 0.25 : 0.0, t % 16 == 0 ? 1.0 : 0.0); SRL_OUT(acc) SRL_OUT(tt) }
    SRL_OUT(rcp(x + 1.0))
    { Masks M; make_masks(L.l, M); SRL_OUT(masked_sum(x, M.le)) SRL_OUT(masked_sum(x, M.ge, 3.0)) }
    {
        double A[ND];
#pragma unroll
        for (int c = 0; c < ND; c++) A[c] = L.arm ? (c == L.l ? 4.0 + L.l : 1.0 / (1.0 + L.l + c)) : 0.0;      // SPD, row l on lane l
        double b = L.arm ? 1.0 + L.l : 0.0, unused = 0.0, B[ND];
#pragma unroll
        for (int c = 0; c < ND; c++) B[c] = A[c];
        gj_step<0, false>(L, B, b);
        SRL_OUT(b)
        gj_step<0, true>(L, A, unused);
#pragma unroll
        for (int c = 0; c < ND; c++) SRL_OUT(A[c])
    }
    {
        GState g; Env e = {};
        g.q = L.arm ? q7[L.l] : 0.0; g.qd = 0.0;
        grefresh<false>(L, g, e);
#pragma unroll
        for (int c = 0; c < 9; c++) SRL_OUT(g.R[c])
#pragma unroll
        for (int c = 0; c < 3; c++) SRL_OUT(g.p[c])
#pragma unroll
        for (int c = 0; c < 3; c++) SRL_OUT(e.grip[c])
    }
#undef SRL_OUT
}


}  // namespace

SRL_GROUP_LAUNCHER(kuka_group_launch_table, true)

int kuka_group_reset_table(Handle *h, const KukaParams &p, const uint8_t *d_mask, const double *d_host_rand, int stride, float *obs) {
    dim3 ggrid((h->n + kGroupEnvs - 1) / kGroupEnvs), gblock(kGroupBlock);
    const bool joints = !h->cfg.is_discrete && h->cfg.action_joints;
#define SRL_GRESET(MODE)                                                                                                                       \
    if (joints) hipLaunchKernelGGL((kuka_group_reset_k<MODE, true>), ggrid, gblock, 0, h->stream, p, *h->kuka, h->rng, h->stats, d_mask, d_host_rand, stride, obs); \
    else hipLaunchKernelGGL((kuka_group_reset_k<MODE, false>), ggrid, gblock, 0, h->stream, p, *h->kuka, h->rng, h->stats, d_mask, d_host_rand, stride, obs)
    switch (h->cfg.rng_mode) {
        case SRLHIP_RNG_HOST: SRL_GRESET(SRLHIP_RNG_HOST); break;
        case SRLHIP_RNG_PHILOX: SRL_GRESET(SRLHIP_RNG_PHILOX); break;
        default: SRL_GRESET(SRLHIP_RNG_MT19937);
    }
#undef SRL_GRESET
    SRL_HIP_CHECK(h, hipGetLastError());
    return 0;
}

int kuka_group_settle_table(Handle *h, const KukaParams &p) {
    hipLaunchKernelGGL(kuka_group_settle_k, dim3(1), dim3(kGroupBlock), 0, h->stream, p, *h->kuka);
    SRL_HIP_CHECK(h, hipGetLastError());
    return 0;
}

int kuka_group_probe(const double *q7_host, double *out_host, int out_doubles) {
    if (out_doubles < kProbeRows * 64) return SRLHIP_EINVAL;
    double *dq = nullptr, *dout = nullptr;
    if (hipMalloc(&dq, 7 * sizeof(double)) != hipSuccess || hipMalloc(&dout, kProbeRows * 64 * sizeof(double)) != hipSuccess) return SRLHIP_ENOMEM;
    (void)hipMemcpy(dq, q7_host, 7 * sizeof(double), hipMemcpyHostToDevice);
    (void)hipMemset(dout, 0, kProbeRows * 64 * sizeof(double));
    hipLaunchKernelGGL(kuka_group_probe_k, dim3(1), dim3(64), 0, 0, dq, dout);
    const hipError_t err = hipDeviceSynchronize();
    (void)hipMemcpy(out_host, dout, kProbeRows * 64 * sizeof(double), hipMemcpyDeviceToHost);
    (void)hipFree(dq); (void)hipFree(dout);
    return err == hipSuccess ? 0 : SRLHIP_EHIP;
}


}  // namespace srl
