"""Endurance: 300 000 steps of 4096 envs on the launching path (early completion signal) and under persistent stepping, same actions;
final state, episode statistics and a running checksum of every reward / done plane must agree (bit-identical paths), no step may
time out.  ~1 minute on the GPU box."""
import sys, time, zlib
import numpy as np
sys.path.insert(0, "robotics-rl-srl_amd")
from srlhip import _lib
N, STEPS = 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 300000
cfg = _lib.default_config(_lib.ENV_KUKA_BUTTON)
cfg.num_envs, cfg.seed0, cfg.rng_mode, cfg.info_bits = N, 77, _lib.RNG_MT19937, 1
res = {}
for name, persistent in (("launching", False), ("persistent", True)):
    h = _lib.Handle(cfg)
    h.reset()
    if persistent:
        h.set_persistent(True)
    rs = np.random.RandomState(5)
    acts = rs.randint(6, size=(4096, N)).astype(np.int32)
    out = (h.new_obs(), np.zeros(N, np.float32), np.zeros(N, np.uint8))
    crc, worst = 0, 0.0
    t0 = time.time()
    for t in range(STEPS):
        t1 = time.perf_counter()
        h.step(acts[t & 4095], out=out)
        dt = time.perf_counter() - t1
        worst = max(worst, dt)
        crc = zlib.crc32(out[2].tobytes(), zlib.crc32(out[1].tobytes(), crc))
        if persistent and t % 50000 == 49999:
            h.get_state(_lib.F_KUKA_Q)                    # park + restart
    wall = time.time() - t0
    res[name] = (crc, h.get_state(_lib.F_KUKA_Q), h.get_state(_lib.F_KUKA_QD), h.episode_stats())
    print("%s: %d steps in %.1f s (%.1f us per step), slowest step %.0f us, checksum %08x, episodes %d" % (
        name, STEPS, wall, wall / STEPS * 1e6, worst * 1e6, crc, int(res[name][3][2].sum())), flush=True)
    h.close()
a, b = res["launching"], res["persistent"]
ok = a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and all(np.array_equal(x, y) for x, y in zip(a[3], b[3]))
print("identical:", ok)
sys.exit(0 if ok else 1)
