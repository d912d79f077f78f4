import sys, numpy as np
sys.path.insert(0, "robotics-rl-srl_amd")
from srlhip.vec_env import HipVecEnv
from srlhip import _lib
n = 256
kw = {"srl_model": "ground_truth"}
a = HipVecEnv("KukaButtonGymEnv-v0", n, seed=3, env_kwargs=kw)
b = HipVecEnv("KukaButtonGymEnv-v0", n, seed=3, env_kwargs=kw, persistent=True)
a.reset(); b.reset()
rs = np.random.RandomState(0)
shown = 0
for t in range(400):
    act = rs.randint(6, size=n)
    act[rs.rand(n) < 0.3] = 4
    a.step(act); b.step(act)
    for f, nm in ((_lib.F_KUKA_Q, "q"), (_lib.F_KUKA_QD, "qd"), (_lib.F_KUKA_BUTTON_Q, "button q/qd"), (_lib.F_KUKA_EE_TARGET, "ee"), (_lib.F_KUKA_GRIPPER, "gripper")):
        qa, qb = a._h.get_state(f), b._h.get_state(f)
        if not np.array_equal(qa, qb):
            qa = qa.reshape(-1, n) if qa.ndim > 1 else qa.reshape(1, n); qb = qb.reshape(qa.shape); d = np.nonzero((qa != qb).any(0))[0]
            print("t", t, nm, "envs differing", len(d), d[:6], "max abs diff", np.abs(qa - qb).max(), "joints", np.nonzero((qa != qb).any(1))[0])
            shown += 1
    if shown >= 4: break
print("done", t)
