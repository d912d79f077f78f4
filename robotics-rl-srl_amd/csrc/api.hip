// api.hip — C-ABI front end of libsrlhip (include/srlhip.h): handle lifetime,
// seeding, host<->device staging, state access, timing.  The env kernels live
// in mobile.hip / kuka.hip.
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <mutex>
#include <cmath>
#include <new>

#include "internal.hpp"

using namespace srl;

namespace {

thread_local std::string g_create_error;
constexpr int kBlock = 256;

bool is_mobile(int kind) { return kind >= SRLHIP_ENV_MOBILE && kind <= SRLHIP_ENV_MOBILE_LINE; }

// RandomState.seed(digits) for the selected envs, one lane per env.
__global__ void mt_seed_k(Mt19937View v, int n, const uint8_t *mask, const uint32_t *digits, const int32_t *len) {
    int e = blockIdx.x * kBlock + threadIdx.x;
    if (e >= n || (mask && !mask[e])) return;
    uint32_t key[2] = {digits[e], digits[n + e]};
    Mt19937 m;
    m.load(v, e);
    m.seed_by_array(key, len[e]);
    m.store(v, e);
}

__global__ void key_seed_k(RngState r, int n, const uint8_t *mask, const uint32_t *lohi) {
    int e = blockIdx.x * kBlock + threadIdx.x;
    if (e >= n || (mask && !mask[e])) return;
    r.key[e] = lohi[e]; r.key[n + e] = lohi[n + e];
    r.ctr[e] = 0; r.act_ctr[e] = 0;
}

int ensure(Handle *h, void **buf, size_t *cap, size_t bytes) {
    if (*cap >= bytes) return 0;
    if (*buf) (void)hipFree(*buf);
    *buf = nullptr; *cap = 0;
    SRL_HIP_CHECK(h, hipMalloc(buf, bytes));
    *cap = bytes;
    return 0;
}
int ensure_pinned(Handle *h, void **buf, size_t *cap, size_t bytes) {
    if (*cap >= bytes) return 0;
    if (*buf) (void)hipHostFree(*buf);
    *buf = nullptr; *cap = 0;
    // mapped: kernels may address it directly (zero-copy step); coherent (fine-grained): what a RESIDENT kernel writes and releases at
    // system scope is visible to the host while the kernel is still running (persistent stepping; + 16: its copy works in dwords)
    SRL_HIP_CHECK(h, hipHostMalloc(buf, bytes + 16, hipHostMallocMapped | hipHostMallocCoherent));
    *cap = bytes;
    return 0;
}
int stage_in(Handle *h, void **buf, size_t *cap, const void *host, size_t bytes) {
    int rc = ensure(h, buf, cap, bytes);
    if (rc) return rc;
    SRL_HIP_CHECK(h, hipMemcpyAsync(*buf, host, bytes, hipMemcpyHostToDevice, h->stream));
    return 0;
}

size_t action_bytes(const Handle *h) {
    return (size_t)h->n * (h->cfg.is_discrete ? sizeof(int32_t) : sizeof(float) * action_dim_of(h->cfg));
}
size_t obs_bytes_per_env(const Handle *h) {
    if (h->cfg.obs_mode == SRLHIP_OBS_RAW_PIXELS)
        return (size_t)h->cfg.img_h * h->cfg.img_w * (h->cfg.multi_view ? 6 : 3);
    return sizeof(float) * obs_dim_of(h->cfg);
}

// ---- persistent stepping: host side of the protocol (internal.hpp PersistArgs) ---------------------------------------------------------
struct PersistHost { volatile uint32_t seq, stop, parked; volatile uint32_t pad[13]; volatile uint32_t done[16]; };     // mapped, coherent; done[0..7]: persistent stepping, one per eighth of the workgroups;
                                                                                                                          // done[8..15]: the early completion signal of single-step launches
PersistHost *persist_ctl(Handle *h) { return static_cast<PersistHost *>(h->persist_host); }

// Resident kernels of several handles on ONE device (the shards of HipVecEnv(device_ids=[0, 0])) must all fit at once: a workgroup that
// waits for the host never makes room for another kernel's.  Per-device tally of the grids of the handles in persistent mode, process-wide.
std::mutex g_persist_mu;
int g_persist_reserved[64] = {0};
void persist_release(Handle *h) {
    if (!h->persist_reserved) return;
    std::lock_guard<std::mutex> lk(g_persist_mu);
    g_persist_reserved[h->cfg.device_id & 63] -= h->persist_reserved;
    h->persist_reserved = 0;
}

// the mapped control block + the device words (relay, arrival counters, start barrier, the single-step signal's counters, timeline stamps)
// shared by persistent stepping and by the early completion signal of single-step launches
int ensure_signal_buffers(Handle *h) {
    if (h->persist_host) return 0;
    const size_t bytes = sizeof(PersistHost), blocks = ((size_t)h->n + 3) / 4;
    if (hipHostMalloc(&h->persist_host, bytes, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) return h->fail(SRLHIP_ENOMEM, "step: hipHostMalloc (control block) failed");
    memset(h->persist_host, 0, bytes);
    const size_t words = 28 * kPersistWordStride + 16 * ((blocks + 7) / 8 * 8);
    if (hipMalloc(reinterpret_cast<void **>(&h->persist_relay), words * sizeof(uint32_t)) != hipSuccess) return h->fail(SRLHIP_ENOMEM, "step: hipMalloc (control words) failed");
    SRL_HIP_CHECK(h, hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(h->persist_relay), 0, words, h->stream));
    return 0;
}

// the resident kernel writes the state back and exits; afterwards the handle is an ordinary one again
int persist_park(Handle *h) {
    if (!h->persist_running) return 0;
    PersistHost *c = persist_ctl(h);
    __atomic_store_n(&c->stop, 1u, __ATOMIC_RELEASE);
    SRL_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    c->stop = 0; c->parked = 0;
    h->persist_running = false;
    return 0;
}

// step_pair: the caller is srlhip_step / _async / _wait.  Every OTHER entry point sees the state in HBM: the resident kernel parks first.
int set_device(Handle *h, bool step_pair = false) {
    SRL_HIP_CHECK(h, hipSetDevice(h->cfg.device_id));
    if (h->persist_running && !step_pair) return persist_park(h);
    return 0;
}

// (re)start the resident kernel; it will treat `start_seq` as the last step it has done
int persist_launch(Handle *h, uint32_t start_seq) {
    PersistHost *c = persist_ctl(h);
    c->stop = 0; c->parked = 0;
    for (uint32_t g = 0; g < 8; g++) c->done[g] = start_seq;
    SRL_HIP_CHECK(h, hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(h->persist_relay), (int)start_seq, 8 * kPersistWordStride, h->stream));
    SRL_HIP_CHECK(h, hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(h->persist_relay + 8 * kPersistWordStride), 0, 12 * kPersistWordStride, h->stream));
    void *dctl = nullptr, *din = nullptr, *dout = nullptr;
    SRL_HIP_CHECK(h, hipHostGetDevicePointer(&dctl, h->persist_host, 0));
    SRL_HIP_CHECK(h, hipHostGetDevicePointer(&din, h->pin_in, 0));
    SRL_HIP_CHECK(h, hipHostGetDevicePointer(&dout, h->pin_out, 0));
    uint32_t *w = static_cast<uint32_t *>(dctl);
    PersistArgs pa;
    pa.seq = w; pa.stop = w + 1; pa.parked = w + 2; pa.done = w + 16; pa.relay = h->persist_relay; pa.count = h->persist_relay + 8 * kPersistWordStride; pa.ctrl = h->persist_relay + 16 * kPersistWordStride;
    pa.start_seq = start_seq;
    { const char *v = getenv("SRLHIP_PERSIST_STAGED"); pa.force_staged = v && atoi(v) != 0; }
    pa.spin_limit = h->persist_park_us / 2 + 1;        // one poll of workgroup 0: one 8-byte PCIe read + s_sleep 8, ~2 us
    const size_t n = (size_t)h->n, ob = obs_bytes_per_env(h) * n, out_rew = (ob + 15) & ~(size_t)15, out_done = out_rew + 4 * n;
    pa.stage = static_cast<const uint32_t *>(h->persist_stage); pa.host_out = static_cast<uint32_t *>(dout);
    pa.rew_dw = (uint32_t)(out_rew / 4); pa.done_dw = (uint32_t)(out_done / 4);
    uint8_t *o = static_cast<uint8_t *>(h->persist_stage), *ho = static_cast<uint8_t *>(dout);
    // (the Kuka kernels take the staging copy and switch to the host's planes themselves; the MobileRobot kernel writes the host's planes)
    int rc = is_mobile(h->cfg.env_kind) ? mobile_persist_start(h, din, reinterpret_cast<float *>(ho), reinterpret_cast<float *>(ho + out_rew), ho + out_done, pa)
                                        : kuka_persist_start(h, din, reinterpret_cast<float *>(o), reinterpret_cast<float *>(o + out_rew), o + out_done, pa);
    if (rc) return rc;
    h->persist_running = true;
    return 0;
}

int seed_impl(Handle *h, const uint8_t *mask, const int64_t *seeds) {
    const int n = h->n;
    std::vector<uint32_t> lohi(2 * (size_t)n), digits(2 * (size_t)n);
    std::vector<int32_t> len(n);
    for (int i = 0; i < n; i++) {
        uint64_t s = (uint64_t)seeds[i];
        lohi[i] = (uint32_t)s; lohi[n + i] = (uint32_t)(s >> 32);
        if (h->cfg.rng_mode == SRLHIP_RNG_MT19937 && (!mask || mask[i])) {
            if (seeds[i] < 0) return h->fail(SRLHIP_EINVAL, "seed must be a non-negative integer (gym seeding)");
            uint32_t d[2];
            len[i] = gym_hash_seed(s, d);
            if (len[i] == 0) return h->fail(SRLHIP_EINVAL, "seed hashes to an empty key");
            digits[i] = d[0]; digits[n + i] = d[1];
        }
    }
    int rc;
    const uint8_t *d_mask = nullptr;
    if (mask) {
        if ((rc = stage_in(h, &h->st_mask, &h->st_mask_sz, mask, n))) return rc;
        d_mask = static_cast<const uint8_t *>(h->st_mask);
    }
    dim3 grid((n + kBlock - 1) / kBlock), block(kBlock);
    if ((rc = stage_in(h, &h->st_rand, &h->st_rand_sz, lohi.data(), lohi.size() * 4))) return rc;
    hipLaunchKernelGGL(key_seed_k, grid, block, 0, h->stream, h->rng, n, d_mask,
                       static_cast<const uint32_t *>(h->st_rand));
    SRL_HIP_CHECK(h, hipGetLastError());
    SRL_HIP_CHECK(h, hipStreamSynchronize(h->stream));   // host vectors die at return
    if (h->cfg.rng_mode == SRLHIP_RNG_MT19937) {
        if ((rc = stage_in(h, &h->st_rand, &h->st_rand_sz, digits.data(), digits.size() * 4))) return rc;
        if ((rc = stage_in(h, &h->st_noise, &h->st_noise_sz, len.data(), len.size() * 4))) return rc;
        hipLaunchKernelGGL(mt_seed_k, grid, block, 0, h->stream, h->rng.mt, n, d_mask,
                           static_cast<const uint32_t *>(h->st_rand), static_cast<const int32_t *>(h->st_noise));
        SRL_HIP_CHECK(h, hipGetLastError());
        SRL_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    }
    return 0;
}

// srlhip_episode_stats_device: Monitor's per-env (r, l) of the last finished episode as f32 / i32 planes in caller memory
__global__ void episode_stats_k(EpisodeStats st, int n, float *ret, int32_t *len, int32_t *fin) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    if (ret) ret[e] = (float)st.last_return[e];
    if (len) len[e] = st.last_length[e];
    if (fin) fin[e] = st.n_finished[e];
}

// Host-pointer calls validate what the kernels cannot afford to: the full-model Kuka kernels are compiled with -fno-honor-nans, so a NaN
// or an infinity in a continuous action or in host-drawn noise would propagate through the step unchecked (the reference's
// np.clip / pybullet would raise or saturate).  Device-pointer callers own their buffers' contents.
bool all_finite_f32(const float *p, size_t count) { for (size_t i = 0; i < count; i++) if (!std::isfinite(p[i])) return false; return true; }
bool all_finite_f64(const double *p, size_t count) { for (size_t i = 0; i < count; i++) if (!std::isfinite(p[i])) return false; return true; }

int field_lookup(Handle *h, int field, void **dptr, size_t *elem, int *count) {
    *count = 1;
    switch (field) {
        case SRLHIP_F_LAST_RETURN: *dptr = h->stats.last_return; *elem = 8; return 0;
        case SRLHIP_F_LAST_LENGTH: *dptr = h->stats.last_length; *elem = 4; return 0;
        case SRLHIP_F_N_FINISHED: *dptr = h->stats.n_finished; *elem = 4; return 0;
        case SRLHIP_F_LAST_REWARD: *dptr = h->stats.last_reward; *elem = 8; return 0;
        case SRLHIP_F_EP_RETURN: *dptr = h->stats.ep_return; *elem = 8; return 0;
        case SRLHIP_F_EP_LENGTH: *dptr = h->stats.ep_length; *elem = 4; return 0;
    }
    if (is_mobile(h->cfg.env_kind)) return mobile_field(h, field, dptr, elem, count);
    return kuka_field(h, field, dptr, elem, count);
}

}  // namespace

extern "C" {

int srlhip_abi_version(void) { return SRLHIP_ABI_VERSION; }

int srlhip_default_config(int32_t env_kind, srlhip_config *cfg) {
    if (!cfg || env_kind < SRLHIP_ENV_MOBILE || env_kind > SRLHIP_ENV_LAST) return SRLHIP_EINVAL;
    memset(cfg, 0, sizeof *cfg);
    cfg->struct_size = (int32_t)sizeof *cfg;
    cfg->env_kind = env_kind;
    cfg->num_envs = 1;
    cfg->is_discrete = 1;
    cfg->force_down = 1;
    cfg->action_repeat = 1;
    cfg->obs_mode = SRLHIP_OBS_GROUND_TRUTH;
    cfg->img_h = cfg->img_w = 224;                                  // RENDER_HEIGHT/WIDTH
    cfg->rng_mode = SRLHIP_RNG_MT19937;
    cfg->auto_reset = 1;
    cfg->max_distance = env_kind >= SRLHIP_ENV_KUKA_BUTTON ? 0.8 : 1.6;   // ctor defaults
    if (env_kind == SRLHIP_ENV_KUKA_2BUTTON) { cfg->max_distance = 2.0; cfg->force_down = 0; }   // kuka_2button_gym_env.py:29
    // every Kuka env integrates the full arm + gripper model on the tree lane-group kernel (Kuka2Button: its two-button form)
    cfg->kuka_model = env_kind >= SRLHIP_ENV_KUKA_BUTTON ? SRLHIP_KUKA_MODEL_FULL : SRLHIP_KUKA_MODEL_LUMPED;
    return 0;
}

const char *srlhip_last_error(srlhip_handle hh) {
    if (!hh) return g_create_error.c_str();
    return reinterpret_cast<Handle *>(hh)->err.c_str();
}

int srlhip_create(const srlhip_config *cfg, srlhip_handle *out) {
    if (!cfg || !out) { g_create_error = "create: null argument"; return SRLHIP_EINVAL; }
    *out = nullptr;
    if (cfg->struct_size != (int32_t)sizeof(srlhip_config)) {
        g_create_error = "create: srlhip_config.struct_size mismatch (ABI)"; return SRLHIP_EINVAL;
    }
    if (cfg->num_envs <= 0) { g_create_error = "create: num_envs must be positive"; return SRLHIP_EINVAL; }
    if (cfg->env_kind < SRLHIP_ENV_MOBILE || cfg->env_kind > SRLHIP_ENV_LAST) {
        g_create_error = "create: unknown env_kind"; return SRLHIP_EINVAL;
    }
    if (cfg->rng_mode < SRLHIP_RNG_HOST || cfg->rng_mode > SRLHIP_RNG_MT19937) {
        g_create_error = "create: unknown rng_mode"; return SRLHIP_EINVAL;
    }
    if (cfg->auto_reset && cfg->rng_mode == SRLHIP_RNG_HOST) {
        g_create_error = "create: auto_reset needs a device RNG mode"; return SRLHIP_EINVAL;
    }
    if (!cfg->is_discrete &&
        (cfg->env_kind == SRLHIP_ENV_MOBILE_1D || cfg->env_kind == SRLHIP_ENV_MOBILE_2TARGET)) {
        // mobile_robot_1D_env.py:44, mobile_robot_2target_env.py:128-129 raise ValueError
        g_create_error = "Only discrete actions is supported"; return SRLHIP_ENOTSUP;
    }
    if (is_mobile(cfg->env_kind) && cfg->obs_mode != SRLHIP_OBS_GROUND_TRUTH &&
        cfg->obs_mode != SRLHIP_OBS_RAW_PIXELS) {
        g_create_error = "create: MobileRobot envs support ground_truth / raw_pixels only"; return SRLHIP_EINVAL;
    }
    // every obs_mode: srlhip_render() also serves ground-truth handles (dataset_generator --img-size), and the rasteriser's
    // LDS band holds one image row of at most 2048 pixels
    if (cfg->img_h < 8 || cfg->img_w < 8 || cfg->img_h > 1024 || cfg->img_w > 1024) {
        g_create_error = "create: img_h / img_w must be in [8, 1024]"; return SRLHIP_EINVAL;
    }
    if (cfg->env_kind >= SRLHIP_ENV_KUKA_BUTTON && cfg->kuka_model != SRLHIP_KUKA_MODEL_LUMPED && cfg->kuka_model != SRLHIP_KUKA_MODEL_FULL) {
        g_create_error = "create: unknown kuka_model"; return SRLHIP_EINVAL;
    }
    if (cfg->info_bits != 0 && cfg->info_bits != 1) { g_create_error = "create: info_bits must be 0 or 1"; return SRLHIP_EINVAL; }
    if (cfg->env_kind >= SRLHIP_ENV_KUKA_BUTTON && cfg->action_repeat < 1) {
        g_create_error = "create: action_repeat must be >= 1"; return SRLHIP_EINVAL;
    }
    hipError_t e = hipSetDevice(cfg->device_id);
    if (e != hipSuccess) {
        g_create_error = std::string("create: hipSetDevice failed (no MI355X visible?): ") + hipGetErrorString(e);
        return SRLHIP_EHIP;
    }
    Handle *h = new (std::nothrow) Handle();
    if (!h) { g_create_error = "create: out of host memory"; return SRLHIP_ENOMEM; }
    h->cfg = *cfg; h->n = cfg->num_envs; h->kuka = nullptr;
    h->st_actions = h->st_noise = h->st_obs = h->st_rew = h->st_done = h->st_mask = h->st_rand = nullptr;
    h->st_actions_sz = h->st_noise_sz = h->st_obs_sz = h->st_rew_sz = h->st_done_sz = h->st_mask_sz = h->st_rand_sz = 0;
    h->pin_in = h->pin_out = nullptr; h->pin_in_sz = h->pin_out_sz = 0;
    h->raster_rays[0] = h->raster_rays[1] = nullptr; h->raster_bg[0] = h->raster_bg[1] = nullptr;
    memset(&h->rng, 0, sizeof h->rng); memset(&h->stats, 0, sizeof h->stats); memset(&h->mobile, 0, sizeof h->mobile);
    int rc = 0;
    auto bail = [&](int code) { g_create_error = h->err; srlhip_destroy(reinterpret_cast<srlhip_handle>(h)); return code; };
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) {
        h->stream = nullptr; h->ev_begin = h->ev_end = nullptr; h->err = "create: hipStreamCreate failed";
        return bail(SRLHIP_EHIP);
    }
    (void)hipEventCreate(&h->ev_begin); (void)hipEventCreate(&h->ev_end);
    const size_t n = (size_t)h->n;
    if ((rc = h->dalloc(&h->rng.key, 2 * n)) || (rc = h->dalloc(&h->rng.ctr, n)) || (rc = h->dalloc(&h->rng.act_ctr, n)))
        return bail(rc);
    if (cfg->rng_mode == SRLHIP_RNG_MT19937) {
        h->rng.mt.stride = (int64_t)n;
        if ((rc = h->dalloc(&h->rng.mt.mt, (size_t)MT_N * n)) || (rc = h->dalloc(&h->rng.mt.mti, n)) ||
            (rc = h->dalloc(&h->rng.mt.has_gauss, n)) || (rc = h->dalloc(&h->rng.mt.gauss, n)))
            return bail(rc);
    }
    EpisodeStats &st = h->stats;
    if ((rc = h->dalloc(&st.ep_return, n)) || (rc = h->dalloc(&st.ep_length, n)) || (rc = h->dalloc(&st.n_finished, n)) || (rc = h->dalloc(&st.last_reward, n)))
        return bail(rc);
    if (!cfg->io_device) {
        // Monitor's (r, l) of the last finished episode where the host can read them without a copy: the kernels' exit stores go
        // over PCIe (12 bytes per env and launch), srlhip_episode_records() hands out the host view
        void *dp = nullptr;
        if (hipHostMalloc(&h->ep_host, 12 * n, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess || hipHostGetDevicePointer(&dp, h->ep_host, 0) != hipSuccess) {
            h->err = "create: hipHostMalloc (episode records) failed"; return bail(SRLHIP_ENOMEM);
        }
        memset(h->ep_host, 0, 12 * n);
        st.last_return = static_cast<double *>(dp);
        st.last_length = reinterpret_cast<int32_t *>(static_cast<uint8_t *>(dp) + 8 * n);
    } else if ((rc = h->dalloc(&st.last_return, n)) || (rc = h->dalloc(&st.last_length, n))) return bail(rc);
    rc = is_mobile(cfg->env_kind) ? mobile_alloc(h) : kuka_alloc(h);
    if (rc) return bail(rc);
    std::vector<int64_t> seeds(n);
    for (size_t i = 0; i < n; i++) seeds[i] = cfg->seed0 + cfg->first_env_id + (int64_t)i;   // environments/utils.py:52
    if ((rc = seed_impl(h, nullptr, seeds.data()))) return bail(rc);
    if (hipStreamSynchronize(h->stream) != hipSuccess) { h->err = "create: device initialisation failed"; return bail(SRLHIP_EHIP); }
    *out = reinterpret_cast<srlhip_handle>(h);
    return 0;
}

int srlhip_destroy(srlhip_handle hh) {
    if (!hh) return SRLHIP_EINVAL;
    Handle *h = reinterpret_cast<Handle *>(hh);
    (void)hipSetDevice(h->cfg.device_id);
    h->persist_step = false;
    (void)persist_park(h);
    persist_release(h);
    if (getenv("SRLHIP_DEBUG_SIGNAL")) fprintf(stderr, "srlhip: handle n=%d: %u signalled steps, %u fell back to the stream synchronisation (XCD mismatch), %u timed out\n", h->n, h->signal_steps, h->signal_fallbacks & 0xffffu, h->signal_fallbacks >> 16);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->persist_host) (void)hipHostFree(h->persist_host);
    if (h->persist_relay) (void)hipFree(h->persist_relay);
    if (h->persist_stage) (void)hipFree(h->persist_stage);
    if (h->kuka) kuka_free(h);
    for (void *p : h->allocs) (void)hipFree(p);
    void *st[] = {h->st_actions, h->st_noise, h->st_obs, h->st_rew, h->st_done, h->st_mask, h->st_rand};
    for (void *p : st) if (p) (void)hipFree(p);
    for (void *p : h->act_plane) if (p) (void)hipFree(p);
    if (h->pin_in) (void)hipHostFree(h->pin_in);
    if (h->pin_out) (void)hipHostFree(h->pin_out);
    if (h->ep_host) (void)hipHostFree(h->ep_host);
    if (h->ev_begin) (void)hipEventDestroy(h->ev_begin);
    if (h->ev_end) (void)hipEventDestroy(h->ev_end);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return 0;
}

int srlhip_obs_dim(srlhip_handle hh) { return hh ? obs_dim_of(reinterpret_cast<Handle *>(hh)->cfg) : SRLHIP_EINVAL; }
int srlhip_obs_bytes(srlhip_handle hh) { return hh ? (int)obs_bytes_per_env(reinterpret_cast<Handle *>(hh)) : SRLHIP_EINVAL; }
int srlhip_action_dim(srlhip_handle hh) { return hh ? action_dim_of(reinterpret_cast<Handle *>(hh)->cfg) : SRLHIP_EINVAL; }
int srlhip_num_actions(srlhip_handle hh) { return hh ? num_actions_of(reinterpret_cast<Handle *>(hh)->cfg) : SRLHIP_EINVAL; }

int srlhip_reset_rand_count(srlhip_handle hh) {
    if (!hh) return SRLHIP_EINVAL;
    Handle *h = reinterpret_cast<Handle *>(hh);
    return is_mobile(h->cfg.env_kind) ? mobile_reset_rand_count(h->cfg) : kuka_reset_rand_count(h->cfg);
}

int srlhip_seed(srlhip_handle hh, const uint8_t *mask, const int64_t *seeds) {
    if (!hh || !seeds) return SRLHIP_EINVAL;
    Handle *h = reinterpret_cast<Handle *>(hh);
    int rc = set_device(h);
    h->prefetch_valid = false;                  // the action-stream counters restart
    h->snap_valid = false;
    return rc ? rc : seed_impl(h, mask, seeds);
}

int srlhip_reset(srlhip_handle hh, const uint8_t *mask, const double *host_rand, void *obs_out) {
    if (!hh) return SRLHIP_EINVAL;
    Handle *h = reinterpret_cast<Handle *>(hh);
    int rc = set_device(h);
    if (rc) return rc;
    const int n = h->n;
    const uint8_t *d_mask = nullptr;
    if (mask) {
        if ((rc = stage_in(h, &h->st_mask, &h->st_mask_sz, mask, n))) return rc;
        d_mask = static_cast<const uint8_t *>(h->st_mask);
    }
    const double *d_rand = host_rand;
    void *d_obs = obs_out;
    const size_t ob = obs_bytes_per_env(h) * n;
    if (!h->cfg.io_device) {
        if (host_rand) {
            size_t bytes = sizeof(double) * (size_t)srlhip_reset_rand_count(hh) * n;
            if ((rc = stage_in(h, &h->st_rand, &h->st_rand_sz, host_rand, bytes))) return rc;
            d_rand = static_cast<const double *>(h->st_rand);
        }
        if (obs_out) {
            // unselected rows must keep the caller's contents
            if ((rc = stage_in(h, &h->st_obs, &h->st_obs_sz, obs_out, ob))) return rc;
            d_obs = h->st_obs;
        }
    }
    const bool pixels = h->cfg.obs_mode == SRLHIP_OBS_RAW_PIXELS;
    rc = is_mobile(h->cfg.env_kind) ? mobile_reset(h, d_mask, d_rand, pixels ? nullptr : static_cast<float *>(d_obs))
                                    : kuka_reset(h, d_mask, d_rand, pixels ? nullptr : d_obs);
    if (rc) return rc;
    if (pixels && d_obs && (rc = raster_render(h, d_obs))) return rc;     // every env is (re)drawn: rows of unselected envs too
    if (!h->cfg.io_device) {
        if (obs_out) SRL_HIP_CHECK(h, hipMemcpyAsync(obs_out, d_obs, ob, hipMemcpyDeviceToHost, h->stream));
        SRL_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    } else if (mask) {
        SRL_HIP_CHECK(h, hipStreamSynchronize(h->stream));   // staged mask must stay valid
    }
    return 0;
}

}  // extern "C"

namespace {
// The host-pointer step in two halves (srlhip_step = both, srlhip_step_async / srlhip_step_wait = one each).
// Small steps (ground-truth observations of a few thousand envs: < 1 MiB each way) are ZERO-COPY: the kernel reads the actions from
// and writes obs / reward / done to mapped pinned host memory over PCIe, so a step is one launch + one stream sync (measured ~10 us
// faster per 4096-env step than enqueueing an H2D and a D2H copy).  Larger steps (images, very large batches): one pinned bounce
// buffer each way -> one H2D and one D2H DMA transfer.
struct StepLayout { size_t ob, ab, in_noise, in_total, out_rew, out_done, out_total; bool pixels, zero_copy; };
StepLayout step_layout(const Handle *h) {
    StepLayout L;
    const size_t n = (size_t)h->n;
    L.ob = obs_bytes_per_env(h) * n; L.ab = action_bytes(h);
    L.in_noise = (L.ab + 15) & ~(size_t)15; L.in_total = L.in_noise + sizeof(double) * n;
    L.out_rew = (L.ob + 15) & ~(size_t)15; L.out_done = L.out_rew + 4 * n; L.out_total = L.out_done + n;
    L.pixels = h->cfg.obs_mode == SRLHIP_OBS_RAW_PIXELS;
    static const bool zc_enabled = [] { const char *v = getenv("SRLHIP_ZERO_COPY"); return !v || atoi(v) != 0; }();   // =0: bounce buffers
    L.zero_copy = zc_enabled && !h->cfg.io_device && !L.pixels && L.out_total <= ((size_t)1 << 20);
    return L;
}

// validate + stage the inputs, enqueue the step (and, in bounce mode, the D2H copy of its outputs) on the handle's stream; no wait
int host_step_begin(Handle *h, const void *actions, const double *host_noise, bool want_obs) {
    const int n = h->n;
    const StepLayout L = step_layout(h);
    int rc;
    if (!h->cfg.is_discrete && !all_finite_f32(static_cast<const float *>(actions), L.ab / sizeof(float)))
        return h->fail(SRLHIP_EINVAL, "step: non-finite continuous action");
    if (host_noise && !all_finite_f64(host_noise, (size_t)n)) return h->fail(SRLHIP_EINVAL, "step: non-finite host_noise");
    if ((rc = ensure_pinned(h, &h->pin_in, &h->pin_in_sz, L.in_total)) || (rc = ensure_pinned(h, &h->pin_out, &h->pin_out_sz, L.out_total)))
        return rc;
    if (h->cfg.is_discrete) {
        // copy and range-check in one pass: the reference indexes a per-action list (`[-dv, dv, 0, 0, 0, 0][action]`) — an IndexError in
        // its worker; -1 is its `None` action.  (Checked here so that the per-step Python path needs no reductions over the batch.)
        const int32_t *src = static_cast<const int32_t *>(actions);
        int32_t *dst = static_cast<int32_t *>(h->pin_in);
        const int32_t top = num_actions_of(h->cfg);
        int32_t bad = 0;
        for (int i = 0; i < n; i++) { const int32_t a = src[i]; dst[i] = a; bad |= (a < -1) | (a >= top); }
        if (bad) return h->fail(SRLHIP_EINVAL, "step: discrete action out of range (valid: -1 = no-op, 0 .. num_actions - 1)");
    } else {
        memcpy(h->pin_in, actions, L.ab);
    }
    if (host_noise) memcpy(static_cast<uint8_t *>(h->pin_in) + L.in_noise, host_noise, sizeof(double) * n);
    uint8_t *din = nullptr, *o = nullptr;
    if (L.zero_copy) {
        void *dp = nullptr;
        SRL_HIP_CHECK(h, hipHostGetDevicePointer(&dp, h->pin_in, 0));
        din = static_cast<uint8_t *>(dp);
        SRL_HIP_CHECK(h, hipHostGetDevicePointer(&dp, h->pin_out, 0));
        o = static_cast<uint8_t *>(dp);
    } else {
        if ((rc = ensure(h, &h->st_actions, &h->st_actions_sz, L.in_total)) || (rc = ensure(h, &h->st_obs, &h->st_obs_sz, L.out_total)))
            return rc;
        SRL_HIP_CHECK(h, hipMemcpyAsync(h->st_actions, h->pin_in, host_noise ? L.in_total : L.ab, hipMemcpyHostToDevice, h->stream));
        din = static_cast<uint8_t *>(h->st_actions);
        o = static_cast<uint8_t *>(h->st_obs);
    }
    if (h->persist_on && L.zero_copy && !host_noise) {
        // persistent stepping: no launch — (re)start the resident kernel if it is parked, then hand it the step's sequence number
        PersistHost *c = persist_ctl(h);
        if (h->persist_running && c->parked) { if ((rc = persist_park(h))) return rc; }
        if (!h->persist_running && (rc = persist_launch(h, h->persist_seq))) return rc;
        h->persist_seq = h->persist_seq + 1 == kPersistPark ? 1 : h->persist_seq + 1;
        __atomic_store_n(&c->seq, h->persist_seq, __ATOMIC_RELEASE);       // the actions above are visible before the number
        h->persist_step = true;                        // host_step_finish collects this step from the resident kernel
        return 0;
    }
    const double *d_noise = host_noise ? reinterpret_cast<const double *>(din + L.in_noise) : nullptr;
    void *d_obs = want_obs ? o : nullptr;
    float *d_rew = reinterpret_cast<float *>(o + L.out_rew);
    uint8_t *d_done = o + L.out_done;
    // Early completion signal (zero-copy steps of the full-model Kuka kernels and of the MobileRobot per-step kernel; SRLHIP_STEP_SIGNAL=0:
    // always wait for the kernel's end):
    // the kernel reports the step's outputs per eighth of its grid (kuka_tree_kernels.hpp), host_step_finish polls those words instead
    // of synchronising the stream — the kernel's exit stores and the completion wake-up leave the step's latency.
    static const bool sig_enabled = [] { const char *v = getenv("SRLHIP_STEP_SIGNAL"); return !v || atoi(v) != 0; }();
    PersistArgs sg{};
    h->signal_wait = false;
    if (sig_enabled && L.zero_copy && !host_noise && !ensure_signal_buffers(h)) {
        if (h->signal_seq >= 0xfffffff0u) {               // the arrival counters count in step with the sequence number: restart both
            SRL_HIP_CHECK(h, hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(h->persist_relay + 20 * kPersistWordStride), 0, 8 * kPersistWordStride, h->stream));
            h->signal_seq = 0;
        }
        void *dctl = nullptr;
        SRL_HIP_CHECK(h, hipHostGetDevicePointer(&dctl, h->persist_host, 0));
        uint32_t *w = static_cast<uint32_t *>(dctl);
        sg.done = w + 16 + 8; sg.count = h->persist_relay + 20 * kPersistWordStride;       // (done[8..15]; counter + XCD tag per eighth)
        sg.start_seq = h->signal_seq + 1;
        h->step_signal = &sg; h->step_signal_armed = false;
    }
    rc = is_mobile(h->cfg.env_kind) ? mobile_step(h, din, d_noise, L.pixels ? nullptr : static_cast<float *>(d_obs), d_rew, d_done)
                                    : kuka_step(h, din, d_noise, L.pixels ? nullptr : d_obs, d_rew, d_done);
    h->step_signal = nullptr;
    if (rc) return rc;
    if (h->step_signal_armed) { h->signal_seq += 1; h->signal_wait = true; h->step_signal_armed = false; }
    if (L.pixels && d_obs && (rc = raster_render(h, d_obs))) return rc;
    if (!L.zero_copy) {
        const size_t from = want_obs ? 0 : L.out_rew;
        SRL_HIP_CHECK(h, hipMemcpyAsync(static_cast<uint8_t *>(h->pin_out) + from, static_cast<uint8_t *>(h->st_obs) + from,
                                        L.out_total - from, hipMemcpyDeviceToHost, h->stream));
    }
    return 0;
}

// wait for the step enqueued by host_step_begin and hand its planes to the caller
int host_step_finish(Handle *h, void *obs_out, float *reward_out, uint8_t *done_out) {
    const StepLayout L = step_layout(h);
    if (h->persist_step) {
        // wait for every workgroup's `done` word; a kernel that parked before it saw this step (its timeout ran out while the caller
        // was busy) is restarted: workgroup 0 relays either the step to everybody or the park token to everybody, never a mix
        PersistHost *c = persist_ctl(h);
        const uint32_t want = h->persist_seq;
        h->persist_step = false;
        uint64_t spins = 0;
        const auto t0 = std::chrono::steady_clock::now();
        // (eighth g reports iff it holds a real workgroup: persist_eighths, set with the mode)
        for (uint32_t b = 0; b < 8; b++) {
            while (((h->persist_eighths >> b) & 1u) && __atomic_load_n(&c->done[b], __ATOMIC_ACQUIRE) != want) {
                if ((++spins & 1023u) != 0) continue;
                if (c->parked) {
                    int rc = persist_park(h);
                    if (rc) return rc;
                    if ((rc = persist_launch(h, want == 1 ? kPersistPark - 1 : want - 1))) return rc;
                    b = 0;
                } else if ((spins & ((1u << 20) - 1)) == 0) {
                    // every few ms: a kernel that is gone without having parked has failed; a step never takes 20 s
                    if (hipStreamQuery(h->stream) != hipErrorNotReady) { h->persist_running = false; return h->fail(SRLHIP_EHIP, "persistent step: the resident kernel is gone"); }
                    if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(20)) {
                        (void)persist_park(h);
                        return h->fail(SRLHIP_EHIP, "persistent step: timed out (a workgroup of the resident kernel never started: is the device shared?)");
                    }
                }
            }
        }
    } else if (h->signal_wait) {
        // the early completion signal of a single-step launch: every eighth of the grid that holds a real workgroup reports the step's
        // sequence number.  Nothing depends on the signal ARRIVING: after ~2 ms without it (or when a workgroup found itself on another
        // XCD than its eighth's) the stream synchronisation below is the answer, as before.
        h->signal_wait = false;
        PersistHost *c = persist_ctl(h);
        const uint32_t want = h->signal_seq;
        bool seen = true;
        uint64_t spins = 0;
        const auto t0 = std::chrono::steady_clock::now();
        bool split = false;                              // an eighth of the grid reported from more than one XCD (~want)
        for (uint32_t b = 0; seen && b < 8; b++)
            for (; (h->signal_eighths >> b) & 1u;) {
                const uint32_t v = __atomic_load_n(&c->done[8 + b], __ATOMIC_ACQUIRE);
                if (v == want) break;
                if (v == ~want) { split = true; break; }
                if ((++spins & 4095u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) { seen = false; break; }
            }
        if (!seen || split) {
            h->signal_fallbacks += 1 + (seen ? 0 : 0x10000);
            SRL_HIP_CHECK(h, hipStreamSynchronize(h->stream));
        }
        h->signal_steps++;
    } else
    SRL_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    const uint8_t *po = static_cast<const uint8_t *>(h->pin_out);
    if (obs_out) memcpy(obs_out, po, L.ob);
    if (reward_out) memcpy(reward_out, po + L.out_rew, 4 * (size_t)h->n);
    if (done_out) memcpy(done_out, po + L.out_done, h->n);
    return 0;
}
}  // namespace

extern "C" {

int srlhip_step(srlhip_handle hh, const void *actions, const double *host_noise, void *obs_out, float *reward_out,
                uint8_t *done_out) {
    if (!hh || !actions) return SRLHIP_EINVAL;
    Handle *h = reinterpret_cast<Handle *>(hh);
    int rc = set_device(h, true);
    if (rc) return rc;
    if (h->cfg.rng_mode == SRLHIP_RNG_HOST && !host_noise)
        return h->fail(SRLHIP_EINVAL, "step: RNG_HOST needs host_noise");
    if (h->step_pending) return h->fail(SRLHIP_EINVAL, "step: a srlhip_step_async is pending (call srlhip_step_wait first)");
    if (!h->cfg.io_device) {
        if ((rc = host_step_begin(h, actions, host_noise, obs_out != nullptr))) return rc;
        return host_step_finish(h, obs_out, reward_out, done_out);
    }
    const bool pixels = h->cfg.obs_mode == SRLHIP_OBS_RAW_PIXELS;
    rc = is_mobile(h->cfg.env_kind) ? mobile_step(h, actions, host_noise, pixels ? nullptr : static_cast<float *>(obs_out), reward_out, done_out)
                                    : kuka_step(h, actions, host_noise, pixels ? nullptr : obs_out, reward_out, done_out);
    if (rc) return rc;
    if (pixels && obs_out && (rc = raster_render(h, obs_out))) return rc;
    return 0;
}

int srlhip_step_async(srlhip_handle hh, const void *actions, const double *host_noise) {
    if (!hh || !actions) return SRLHIP_EINVAL;
    Handle *h = reinterpret_cast<Handle *>(hh);
    if (h->cfg.io_device) return h->fail(SRLHIP_EINVAL, "step_async: host-pointer handles only (on a device-pointer handle srlhip_step already only enqueues)");
    int rc = set_device(h, true);
    if (rc) return rc;
    if (h->cfg.rng_mode == SRLHIP_RNG_HOST && !host_noise) return h->fail(SRLHIP_EINVAL, "step_async: RNG_HOST needs host_noise");
    if (h->step_pending) return h->fail(SRLHIP_EINVAL, "step_async: the previous srlhip_step_async was not collected (srlhip_step_wait)");
    if ((rc = host_step_begin(h, actions, host_noise, true))) return rc;
    h->step_pending = true;
    return 0;
}

int srlhip_step_wait(srlhip_handle hh, void *obs_out, float *reward_out, uint8_t *done_out) {
    if (!hh) return SRLHIP_EINVAL;
    Handle *h = reinterpret_cast<Handle *>(hh);
    if (!h->step_pending) return h->fail(SRLHIP_EINVAL, "step_wait: no srlhip_step_async is pending");
    int rc = set_device(h, true);
    if (rc) return rc;
    h->step_pending = false;
    return host_step_finish(h, obs_out, reward_out, done_out);
}

int srlhip_set_persistent(srlhip_handle hh, int32_t on, int32_t park_us) {
    if (!hh) return SRLHIP_EINVAL;
    Handle *h = reinterpret_cast<Handle *>(hh);
    if (h->step_pending) return h->fail(SRLHIP_EINVAL, "set_persistent: a srlhip_step_async is pending");
    int rc = set_device(h);                            // (parks a resident kernel)
    if (rc) return rc;
    if (!on) { h->persist_on = false; persist_release(h); return 0; }
    if (h->cfg.io_device) return h->fail(SRLHIP_ENOTSUP, "set_persistent: host-pointer handles only");
    int capacity = 0;
    uint32_t eighths = 0;
    const bool mobile = is_mobile(h->cfg.env_kind);
    const int blocks = mobile ? mobile_persist_blocks(h, &capacity, &eighths) : kuka_persist_blocks(h, &capacity);
    if (!mobile) {                  // eighth g = the contiguous workgroup range [g, g + 1) * ceil(blocks / 8) (kuka_tree_kernels.hpp)
        const uint32_t per = ((uint32_t)(blocks > 0 ? blocks : 0) + 7) / 8;
        for (uint32_t g = 0; g < 8 && g * per < (uint32_t)(blocks > 0 ? blocks : 0); g++) eighths |= 1u << g;
    }
    if (blocks <= 0 || !step_layout(h).zero_copy)
        return h->fail(SRLHIP_ENOTSUP, "set_persistent: needs a MobileRobot env, KukaButtonGymEnv, KukaMovingButtonGymEnv or Kuka2ButtonGymEnv (full model) on a device RNG mode, non-pixel "
                                       "observations, zero-copy step buffers, and a batch whose wavefronts are all resident at once (4096 envs on an MI355X)");
    if ((rc = ensure_signal_buffers(h))) return rc;
    if (!h->persist_stage) {
        const StepLayout L = step_layout(h);
        if ((rc = ensure_pinned(h, &h->pin_in, &h->pin_in_sz, L.in_total)) || (rc = ensure_pinned(h, &h->pin_out, &h->pin_out_sz, L.out_total))) return rc;
        if (hipMalloc(&h->persist_stage, L.out_total + 16) != hipSuccess) return h->fail(SRLHIP_ENOMEM, "set_persistent: hipMalloc failed");
    }
    if (!h->persist_reserved) {
        const int grid = mobile ? blocks : (blocks + 7) / 8 * 8;
        std::lock_guard<std::mutex> lk(g_persist_mu);
        if (g_persist_reserved[h->cfg.device_id & 63] + grid > capacity)
            return h->fail(SRLHIP_ENOTSUP, "set_persistent: the resident kernels of the handles already in persistent mode on this device leave no room for this one");
        g_persist_reserved[h->cfg.device_id & 63] += grid;
        h->persist_reserved = grid;
    }
    h->persist_blocks = (uint32_t)blocks;
    h->persist_eighths = eighths;
    h->persist_park_us = park_us > 0 ? (uint32_t)park_us : 2000u;
    h->persist_on = true;
    return 0;
}

#if defined(SRL_PERSIST_PROF)
// timeline build only (profiles/probes/persist_timeline.py): parks the resident kernel and returns the 8 stamps of every workgroup's last step
int srlhip_debug_persist_prof(srlhip_handle hh, uint64_t *out, int32_t blocks) {
    Handle *h = reinterpret_cast<Handle *>(hh);
    int rc = set_device(h);
    if (rc) return rc;
    SRL_HIP_CHECK(h, hipMemcpy(out, h->persist_relay + 28 * kPersistWordStride, 64 * (size_t)blocks, hipMemcpyDeviceToHost));
    return 0;
}
#endif

int srlhip_step_pending(srlhip_handle hh) { return hh ? (reinterpret_cast<Handle *>(hh)->step_pending ? 1 : 0) : SRLHIP_EINVAL; }

int srlhip_rollout(srlhip_handle hh, int32_t T, const void *actions_TN, void *obs_TN, float *reward_TN,
                   uint8_t *done_TN, void *act_out_TN) {
    if (!hh || T <= 0) return SRLHIP_EINVAL;
    Handle *h = reinterpret_cast<Handle *>(hh);
    int rc = set_device(h);
    if (rc) return rc;
    if (!h->cfg.auto_reset || h->cfg.rng_mode == SRLHIP_RNG_HOST)
        return h->fail(SRLHIP_EINVAL, "rollout: needs auto_reset and a device RNG mode");
    const size_t n = (size_t)h->n, tn = n * (size_t)T;
    const void *d_act = actions_TN; void *d_obs = obs_TN; float *d_rew = reward_TN; uint8_t *d_done = done_TN;
    void *d_act_out = act_out_TN;
    const size_t ob = obs_bytes_per_env(h) * tn, ab = action_bytes(h) * (size_t)T;
    if (!h->cfg.io_device) {
        if (actions_TN && !h->cfg.is_discrete && !all_finite_f32(static_cast<const float *>(actions_TN), ab / sizeof(float)))
            return h->fail(SRLHIP_EINVAL, "rollout: non-finite continuous action");
        if (actions_TN) { if ((rc = stage_in(h, &h->st_actions, &h->st_actions_sz, actions_TN, ab))) return rc; d_act = h->st_actions; }
        else if (act_out_TN) { if ((rc = ensure(h, &h->st_actions, &h->st_actions_sz, ab))) return rc; d_act_out = h->st_actions; }
        if (obs_TN) { if ((rc = ensure(h, &h->st_obs, &h->st_obs_sz, ob))) return rc; d_obs = h->st_obs; }
        if (reward_TN) { if ((rc = ensure(h, &h->st_rew, &h->st_rew_sz, 4 * tn))) return rc; d_rew = static_cast<float *>(h->st_rew); }
        if (done_TN) { if ((rc = ensure(h, &h->st_done, &h->st_done_sz, tn))) return rc; d_done = static_cast<uint8_t *>(h->st_done); }
    }
    if (h->cfg.obs_mode != SRLHIP_OBS_RAW_PIXELS) {
        rc = is_mobile(h->cfg.env_kind)
                 ? mobile_rollout(h, T, d_act, static_cast<float *>(d_obs), d_rew, d_done, actions_TN ? nullptr : d_act_out)
                 : kuka_rollout(h, T, d_act, d_obs, d_rew, d_done, actions_TN ? nullptr : d_act_out);
        if (rc) return rc;
    } else {
        // images are drawn between steps: one stepper launch + one rasteriser launch per step, same stream
        const size_t a_step = action_bytes(h), o_step = obs_bytes_per_env(h) * n;
        for (int32_t t = 0; t < T; t++) {
            const void *a_t = d_act ? static_cast<const uint8_t *>(d_act) + (size_t)t * a_step : nullptr;
            void *ao_t = (!actions_TN && d_act_out) ? static_cast<uint8_t *>(d_act_out) + (size_t)t * a_step : nullptr;
            float *r_t = d_rew ? d_rew + (size_t)t * n : nullptr;
            uint8_t *dn_t = d_done ? d_done + (size_t)t * n : nullptr;
            rc = is_mobile(h->cfg.env_kind) ? mobile_rollout(h, 1, a_t, nullptr, r_t, dn_t, ao_t)
                                            : kuka_rollout(h, 1, a_t, nullptr, r_t, dn_t, ao_t);
            if (rc) return rc;
            if (d_obs && (rc = raster_render(h, static_cast<uint8_t *>(d_obs) + (size_t)t * o_step))) return rc;
        }
    }
    if (!h->cfg.io_device) {
        if (obs_TN) SRL_HIP_CHECK(h, hipMemcpyAsync(obs_TN, d_obs, ob, hipMemcpyDeviceToHost, h->stream));
        if (reward_TN) SRL_HIP_CHECK(h, hipMemcpyAsync(reward_TN, d_rew, 4 * tn, hipMemcpyDeviceToHost, h->stream));
        if (done_TN) SRL_HIP_CHECK(h, hipMemcpyAsync(done_TN, d_done, tn, hipMemcpyDeviceToHost, h->stream));
        if (act_out_TN && !actions_TN) SRL_HIP_CHECK(h, hipMemcpyAsync(act_out_TN, d_act_out, ab, hipMemcpyDeviceToHost, h->stream));
        SRL_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    }
    return 0;
}

int srlhip_get_state(srlhip_handle hh, int32_t field, void *out) {
    if (!hh || !out) return SRLHIP_EINVAL;
    Handle *h = reinterpret_cast<Handle *>(hh);
    int rc = set_device(h);
    if (rc) return rc;
    void *d; size_t elem; int count;
    if ((rc = field_lookup(h, field, &d, &elem, &count))) return rc;
    SRL_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    if (h->ep_host && (field == SRLHIP_F_LAST_RETURN || field == SRLHIP_F_LAST_LENGTH)) {        // mapped host block (create)
        memcpy(out, static_cast<const uint8_t *>(h->ep_host) + (field == SRLHIP_F_LAST_LENGTH ? 8 * (size_t)h->n : 0), elem * (size_t)h->n);
        return 0;
    }
    SRL_HIP_CHECK(h, hipMemcpy(out, d, elem * count * (size_t)h->n, hipMemcpyDeviceToHost));
    return 0;
}

int srlhip_set_state(srlhip_handle hh, int32_t field, const void *in) {
    if (!hh || !in) return SRLHIP_EINVAL;
    Handle *h = reinterpret_cast<Handle *>(hh);
    int rc = set_device(h);
    if (rc) return rc;
    void *d; size_t elem; int count;
    if ((rc = field_lookup(h, field, &d, &elem, &count))) return rc;
    h->snap_valid = false;                 // (MobileRobot: the rollout's chained snapshot no longer equals the live state)
    SRL_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    if (h->ep_host && (field == SRLHIP_F_LAST_RETURN || field == SRLHIP_F_LAST_LENGTH)) {
        memcpy(static_cast<uint8_t *>(h->ep_host) + (field == SRLHIP_F_LAST_LENGTH ? 8 * (size_t)h->n : 0), in, elem * (size_t)h->n);
        return 0;
    }
    SRL_HIP_CHECK(h, hipMemcpy(d, in, elem * count * (size_t)h->n, hipMemcpyHostToDevice));
    if (field == SRLHIP_F_KUKA_Q || field == SRLHIP_F_KUKA_GRIPPER_Q) return kuka_refresh(h);        // derived planes: sin/cos of q, gripper position
    return 0;
}

int srlhip_device_ptr(srlhip_handle hh, int32_t field, void **dptr) {
    if (!hh || !dptr) return SRLHIP_EINVAL;
    Handle *h = reinterpret_cast<Handle *>(hh);
    size_t elem; int count;
    return field_lookup(h, field, dptr, &elem, &count);
}

int srlhip_render(srlhip_handle hh, void *img_out) {
    if (!hh || !img_out) return SRLHIP_EINVAL;
    Handle *h = reinterpret_cast<Handle *>(hh);
    int rc = set_device(h);
    if (rc) return rc;
    const size_t bytes = (size_t)h->cfg.img_h * h->cfg.img_w * (h->cfg.multi_view ? 6 : 3) * h->n;
    void *d_img = img_out;
    if (!h->cfg.io_device) { if ((rc = ensure(h, &h->st_obs, &h->st_obs_sz, bytes))) return rc; d_img = h->st_obs; }
    if ((rc = raster_render(h, d_img))) return rc;
    if (!h->cfg.io_device) {
        SRL_HIP_CHECK(h, hipMemcpyAsync(img_out, d_img, bytes, hipMemcpyDeviceToHost, h->stream));
        SRL_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    }
    return 0;
}

int srlhip_episode_stats(srlhip_handle hh, double *last_return, int32_t *last_length, int32_t *n_finished) {
    if (!hh) return SRLHIP_EINVAL;
    Handle *h = reinterpret_cast<Handle *>(hh);
    int rc = set_device(h);
    if (rc) return rc;
    SRL_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    const size_t n = (size_t)h->n;
    if (h->ep_host) {          // host-pointer handle: the records already are in host memory
        if (last_return) memcpy(last_return, h->ep_host, 8 * n);
        if (last_length) memcpy(last_length, static_cast<const uint8_t *>(h->ep_host) + 8 * n, 4 * n);
    } else {
        if (last_return) SRL_HIP_CHECK(h, hipMemcpy(last_return, h->stats.last_return, 8 * n, hipMemcpyDeviceToHost));
        if (last_length) SRL_HIP_CHECK(h, hipMemcpy(last_length, h->stats.last_length, 4 * n, hipMemcpyDeviceToHost));
    }
    if (n_finished) SRL_HIP_CHECK(h, hipMemcpy(n_finished, h->stats.n_finished, 4 * n, hipMemcpyDeviceToHost));
    return 0;
}

int srlhip_episode_records(srlhip_handle hh, const double **last_return, const int32_t **last_length) {
    if (!hh) return SRLHIP_EINVAL;
    Handle *h = reinterpret_cast<Handle *>(hh);
    if (!h->ep_host) return h->fail(SRLHIP_EINVAL, "episode_records: host-pointer handles only (cfg.io_device = 0); device-pointer handles use srlhip_episode_stats_device");
    if (last_return) *last_return = static_cast<const double *>(h->ep_host);
    if (last_length) *last_length = reinterpret_cast<const int32_t *>(static_cast<const uint8_t *>(h->ep_host) + 8 * (size_t)h->n);
    return 0;
}

int srlhip_episode_stats_device(srlhip_handle hh, float *d_last_return, int32_t *d_last_length, int32_t *d_n_finished) {
    if (!hh) return SRLHIP_EINVAL;
    Handle *h = reinterpret_cast<Handle *>(hh);
    int rc = set_device(h);
    if (rc) return rc;
    hipLaunchKernelGGL(episode_stats_k, dim3((h->n + 255) / 256), dim3(256), 0, h->stream, h->stats, h->n, d_last_return,
                       d_last_length, d_n_finished);
    SRL_HIP_CHECK(h, hipGetLastError());
    return 0;
}

int srlhip_kuka_default_model(srlhip_kuka_model *m) {
    if (!m) return SRLHIP_EINVAL;
    kuka_default_model(reinterpret_cast<double *>(m));
    return 0;
}

int srlhip_set_kuka_model(srlhip_handle hh, const srlhip_kuka_model *m) {
    if (!hh || !m) return SRLHIP_EINVAL;
    Handle *h = reinterpret_cast<Handle *>(hh);
    if (is_mobile(h->cfg.env_kind)) return h->fail(SRLHIP_EINVAL, "set_kuka_model: not a Kuka handle");
    int rc = set_device(h);
    if (rc) return rc;
    for (int i = 0; i < 7; i++)
        if (!(m->mass[i] > 0.0) || !(m->inertia[i][0] > 0.0) || !(m->inertia[i][1] > 0.0) || !(m->inertia[i][2] > 0.0) || !(m->joint_lower[i] < m->joint_upper[i]))
            return h->fail(SRLHIP_EINVAL, "set_kuka_model: masses and principal inertias must be positive, joint_lower < joint_upper");
    return kuka_set_model(h, reinterpret_cast<const double *>(m));
}

int srlhip_kuka_tree_default_model(srlhip_kuka_tree_model *m) {
    if (!m) return SRLHIP_EINVAL;
    kuka_default_tree_model(reinterpret_cast<double *>(m));
    return 0;
}

int srlhip_set_kuka_tree_model(srlhip_handle hh, const srlhip_kuka_tree_model *m) {
    if (!hh || !m) return SRLHIP_EINVAL;
    Handle *h = reinterpret_cast<Handle *>(hh);
    if (is_mobile(h->cfg.env_kind)) return h->fail(SRLHIP_EINVAL, "set_kuka_tree_model: not a Kuka handle");
    if (h->cfg.kuka_model != SRLHIP_KUKA_MODEL_FULL) return h->fail(SRLHIP_EINVAL, "set_kuka_tree_model: the handle was created with the lumped model");
    const int nd = (int)m->nd, ns = (int)m->nsphere;
    if (nd != 12 || ns < 0 || ns > 16 || (int)m->max_generic_rows < 0 || (int)m->max_generic_rows > 6 || (int)m->ee_link != 6 || (int)m->grip_link < 0 || (int)m->grip_link >= nd)
        return h->fail(SRLHIP_EINVAL, "set_kuka_tree_model: 12 DoFs, <= 16 spheres, end effector = link 6, max_generic_rows <= 6 (the lane group's row bank)");
    for (int i = 0; i < nd; i++) {
        const srlhip_kuka_tree_joint &J = m->j[i];
        const double a2 = J.axis[0] * J.axis[0] + J.axis[1] * J.axis[1] + J.axis[2] * J.axis[2];
        if (!((int)J.parent < i && (int)J.parent >= -1) || !(J.mass > 0) || !(J.inertia[0] > 0 && J.inertia[3] > 0 && J.inertia[5] > 0) ||
            !(a2 > 0.999999 && a2 < 1.000001) || !(J.max_force > 0) || !(J.kp > 0))
            return h->fail(SRLHIP_EINVAL, "set_kuka_tree_model: parents before children, positive masses / inertias / motor gains, unit axes");
        if (i < 7 && ((int)J.parent != i - 1)) return h->fail(SRLHIP_EINVAL, "set_kuka_tree_model: DoFs 0..6 must be the serial arm");
    }
    const int detail = (int)m->solver_detail;
    if (m->solver_detail != (double)detail || detail < 0 || detail > (SRLHIP_KUKA_DETAIL_ALT_SWEEP | SRLHIP_KUKA_DETAIL_BODY_ORDER | SRLHIP_KUKA_DETAIL_FRICTION2) ||
        !(m->contact_erp >= 0.0 && m->contact_erp <= 1.0) || !(m->limit_erp >= 0.0 && m->limit_erp <= 1.0) || !(m->linear_slop >= 0.0 && m->linear_slop <= 0.01))
        return h->fail(SRLHIP_EINVAL, "set_kuka_tree_model: solver_detail is a mask of SRLHIP_KUKA_DETAIL_* bits, erp values in [0, 1], linear_slop in [0, 0.01]");
    for (int k = 0; k < ns; k++) if ((int)m->s[k].link < 0 || (int)m->s[k].link >= nd || !(m->s[k].r > 0)) return h->fail(SRLHIP_EINVAL, "set_kuka_tree_model: bad sphere");
    return kuka_set_tree_model(h, reinterpret_cast<const double *>(m));
}

int srlhip_kuka_kernel(srlhip_handle hh) {
    if (!hh) return SRLHIP_EINVAL;
    Handle *h = reinterpret_cast<Handle *>(hh);
    if (is_mobile(h->cfg.env_kind)) return SRLHIP_EINVAL;
    return kuka_uses_group_kernel(h);
}

int srlhip_selftest_group_primitives(int32_t device_id, const double *q7, double *out, int32_t out_doubles) {
    if (!q7 || !out) return SRLHIP_EINVAL;
    if (hipSetDevice(device_id) != hipSuccess) return SRLHIP_EHIP;
    return kuka_group_probe(q7, out, out_doubles);
}

int srlhip_sync(srlhip_handle hh) {
    if (!hh) return SRLHIP_EINVAL;
    Handle *h = reinterpret_cast<Handle *>(hh);
    SRL_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    return 0;
}

int srlhip_copy_async(srlhip_handle hh, void *dst, const void *src, size_t bytes) {
    if (!hh || (bytes && (!dst || !src))) return SRLHIP_EINVAL;
    Handle *h = reinterpret_cast<Handle *>(hh);
    int rc = set_device(h);
    if (rc) return rc;
    if (bytes) SRL_HIP_CHECK(h, hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, h->stream));
    return 0;
}

int srlhip_stream(srlhip_handle hh, void **hip_stream) {
    if (!hh || !hip_stream) return SRLHIP_EINVAL;
    *hip_stream = reinterpret_cast<Handle *>(hh)->stream;
    return 0;
}

// ---- HIP graphs: capture whatever the caller enqueues on the handle's stream between begin and end (device-pointer
// calls of this library, srlhip_encoder_forward on the same stream, ...) and replay it with one launch per step.
struct srlhip_graph { hipGraph_t graph; hipGraphExec_t exec; };

int srlhip_graph_begin(srlhip_handle hh) {
    if (!hh) return SRLHIP_EINVAL;
    Handle *h = reinterpret_cast<Handle *>(hh);
    if (!h->cfg.io_device) return h->fail(SRLHIP_EINVAL, "graph capture needs io_device = 1 (host-pointer calls synchronise)");
    int rc = set_device(h);
    if (rc) return rc;
    h->prefetch_valid = false;             // a captured synthetic-agent rollout moves the action-stream counters on the device at every replay
    h->snap_valid = false;
    SRL_HIP_CHECK(h, hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
    return 0;
}

int srlhip_graph_end(srlhip_handle hh, srlhip_graph_handle *out) {
    if (!hh || !out) return SRLHIP_EINVAL;
    Handle *h = reinterpret_cast<Handle *>(hh);
    *out = nullptr;
    hipGraph_t g = nullptr;
    SRL_HIP_CHECK(h, hipStreamEndCapture(h->stream, &g));
    hipGraphExec_t ex = nullptr;
    hipError_t e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
    if (e != hipSuccess) {
        (void)hipGraphDestroy(g);
        return h->fail(SRLHIP_EHIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(e));
    }
    srlhip_graph *sg = new (std::nothrow) srlhip_graph();
    if (!sg) { (void)hipGraphExecDestroy(ex); (void)hipGraphDestroy(g); return h->fail(SRLHIP_ENOMEM, "graph handle"); }
    sg->graph = g; sg->exec = ex;
    *out = sg;
    return 0;
}

int srlhip_graph_launch(srlhip_handle hh, srlhip_graph_handle g) {
    if (!hh || !g) return SRLHIP_EINVAL;
    Handle *h = reinterpret_cast<Handle *>(hh);
    h->prefetch_valid = false;             // (see srlhip_graph_begin: the action plane drawn ahead belongs to counters the replay has consumed)
    h->snap_valid = false;                 // ... and the replay moved the live state
    SRL_HIP_CHECK(h, hipGraphLaunch(g->exec, h->stream));
    return 0;
}

int srlhip_graph_destroy(srlhip_graph_handle g) {
    if (!g) return SRLHIP_OK;
    (void)hipGraphExecDestroy(g->exec);
    (void)hipGraphDestroy(g->graph);
    delete g;
    return SRLHIP_OK;
}

int srlhip_timing_begin(srlhip_handle hh) {
    if (!hh) return SRLHIP_EINVAL;
    Handle *h = reinterpret_cast<Handle *>(hh);
    SRL_HIP_CHECK(h, hipEventRecord(h->ev_begin, h->stream));
    return 0;
}

int srlhip_timing_end(srlhip_handle hh, float *elapsed_ms) {
    if (!hh || !elapsed_ms) return SRLHIP_EINVAL;
    Handle *h = reinterpret_cast<Handle *>(hh);
    SRL_HIP_CHECK(h, hipEventRecord(h->ev_end, h->stream));
    SRL_HIP_CHECK(h, hipEventSynchronize(h->ev_end));
    SRL_HIP_CHECK(h, hipEventElapsedTime(elapsed_ms, h->ev_begin, h->ev_end));
    return 0;
}

}  // extern "C"
