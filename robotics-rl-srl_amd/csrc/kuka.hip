// kuka.hip — placeholder until the Kuka stepper lands (next milestone).
#include "internal.hpp"
namespace srl {
struct KukaState {};
int kuka_alloc(Handle *h) { return h->fail(SRLHIP_ENOTSUP, "KukaButtonGymEnv kernels not built yet"); }
void kuka_free(Handle *) {}
int kuka_reset(Handle *h, const uint8_t *, const double *, void *) { return h->fail(SRLHIP_ENOTSUP, "kuka"); }
int kuka_step(Handle *h, const void *, const double *, void *, float *, uint8_t *) { return h->fail(SRLHIP_ENOTSUP, "kuka"); }
int kuka_rollout(Handle *h, int, const void *, void *, float *, uint8_t *, void *) { return h->fail(SRLHIP_ENOTSUP, "kuka"); }
int kuka_field(Handle *h, int, void **, size_t *, int *) { return h->fail(SRLHIP_EINVAL, "unknown field"); }
int kuka_reset_rand_count(const srlhip_config &) { return 0; }
}  // namespace srl
