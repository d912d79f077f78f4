"""Timeline of one persistent step (srlhip_set_persistent), from device-side stamps: build the probe library with
   make -C robotics-rl-srl_amd/csrc persistprof   and run   SRLHIP_LIB=.../build/libsrlhip_pprof.so python profiles/probes/persist_timeline.py [n]
Stamps per workgroup (100 MHz device clock), relative to workgroup 0's token: 0 token seen, 1 actions loaded, 2 env step done,
3 staging stores written through, 4 arrival counted, 5 (copier only) planes copied, 6 (copier) release fence done."""
import ctypes, sys, time
import numpy as np
sys.path.insert(0, "robotics-rl-srl_amd")
from srlhip import _lib
from srlhip.vec_env import HipVecEnv

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
env = HipVecEnv("KukaButtonGymEnv-v0", n, seed=3, env_kwargs={"srl_model": "ground_truth"}, persistent=True, park_us=3000)
env.reset()
rs = np.random.RandomState(0)
for t in range(1100):
    env.step(rs.randint(6, size=n))
lib = _lib.load()
blocks = (n + 3) // 4
buf = np.zeros((blocks, 8), np.uint64)
names = ["token seen", "actions loaded", "env step done", "stores written through", "arrival counted", "copied (copiers)", "fence done (copiers)"]
for rep in range(6):
    ts = []
    for t in range(20):
        a = rs.randint(6, size=n)
        t0 = time.perf_counter(); env.step(a); ts.append(time.perf_counter() - t0)
    h = env._shards[0].h
    assert lib.srlhip_debug_persist_prof(h._h, buf.ctypes.data_as(ctypes.c_void_p), blocks) == 0
    rel = (buf.astype(np.int64) - int(buf[0, 0])) / 100.0          # us
    print("step (host, last of 20): %.1f us" % (ts[-1] * 1e6))
    for k, nm in enumerate(names):
        col = rel[:, k]
        if k >= 5:
            col = col[(buf[:, k] >= buf[:, 0].min())]                 # only copiers of THIS step stamped these
        if len(col):
            print("   %-26s min %6.2f  median %6.2f  p99 %6.2f  max %6.2f us   (n=%d)" % (nm, col.min(), np.median(col), np.percentile(col, 99), col.max(), len(col)))
env.close()
