import sys, time, numpy as np
sys.path.insert(0, "robotics-rl-srl_amd")
from srlhip.vec_env import HipVecEnv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
kw = {"srl_model": "ground_truth"}
a = HipVecEnv("KukaButtonGymEnv-v0", n, seed=3, env_kwargs=kw)
b = HipVecEnv("KukaButtonGymEnv-v0", n, seed=3, env_kwargs=kw, persistent=True, park_us=3000)
print("reset equal", np.array_equal(a.reset(), b.reset()))
rs = np.random.RandomState(0)
bad = [0, 0, 0]
for t in range(300):
    act = rs.randint(6, size=n)
    x, y = a.step(act), b.step(act)
    for k in range(3):
        if not np.array_equal(x[k], y[k]):
            bad[k] += 1
            if bad[k] <= 2:
                d = np.nonzero((x[k] != y[k]).reshape(n, -1).any(1))[0]
                print("t", t, "plane", k, "envs differing", len(d), d[:8], x[k][d[:2]], y[k][d[:2]])
print("steps with mismatching obs/rew/done:", bad)
for name, env in (("launch", a), ("resident", b)):
    ts = []
    for t in range(600):
        act = rs.randint(6, size=n)
        t0 = time.perf_counter(); env.step(act); ts.append(time.perf_counter() - t0)
    ts = np.array(ts[100:]) * 1e6
    print(name, "median %.1f p10 %.1f p90 %.1f max %.1f us" % (np.median(ts), np.percentile(ts, 10), np.percentile(ts, 90), ts.max()))
a.close(); b.close()
