"""__graft_entry__.smoke(), Kuka leg: one small KukaButtonGymEnv rollout on cuda:0 vs the oracle."""
import numpy as np


def run():
    from oracle import kuka_clib
    from srlhip import _lib
    n, T = 64, 450
    cfg = _lib.default_config(_lib.ENV_KUKA_BUTTON)
    cfg.num_envs, cfg.seed0 = n, 1
    h = _lib.Handle(cfg)
    actions = np.random.RandomState(0).randint(6, size=(T, n)).astype(np.int32)
    obs0 = h.reset()
    out = h.rollout(T, actions=actions)
    kuka_clib.set_full(cfg.kuka_model == _lib.KUKA_MODEL_FULL)       # the oracle integrates the model the handle integrates (default: full gripper)
    try:
        ora = kuka_clib.rollout(1 + np.arange(n), T, actions=actions, trace=False)
    finally:
        kuka_clib.set_full(False)
    q_err = np.abs(h.get_state(_lib.F_KUKA_Q).T - ora["final_state"][:, :7]).max()
    assert np.abs(obs0 - ora["obs0"]).max() <= 1e-4 and np.abs(out["obs"] - ora["obs"]).max() <= 1e-4
    assert q_err <= 1e-4, q_err
    assert np.array_equal(out["reward"], ora["reward"]) and np.array_equal(out["done"], ora["done"])
    h_name, kernel = h.kuka_model_name(), h.kuka_kernel()
    h.close()
    print("smoke: KukaButtonGymEnv-v0 ({}, kernel {}) x{} envs x{} steps: max|q-q_oracle|={:.2e}, reward/done bit-exact".format(
        h_name, kernel, n, T, q_err))
