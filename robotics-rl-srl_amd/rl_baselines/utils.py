"""rl_baselines/utils.py:194-229 — createEnvs with the batched GPU vec env spliced in
where the reference builds `[makeEnv(...)] -> DummyVecEnv | SubprocVecEnv` (:213-220).
Everything after that line is unchanged: VecFrameStack, then VecNormalize for non-pixel
observations."""
from environments import ThreadingType

try:                                        # pragma: no cover - stable_baselines is absent here
    from stable_baselines.common.vec_env import VecNormalize, VecFrameStack
except Exception:                           # noqa: BLE001
    from srlhip.vec_wrappers import VecNormalize, VecFrameStack
from srlhip.vec_env import HipVecEnv


def createEnvs(args, allow_early_resets=False, env_kwargs=None, load_path_normalise=None):
    from environments.registry import registered_env
    assert not (registered_env[args.env][3] is ThreadingType.NONE and args.num_cpu != 1), \
        "Error: cannot have more than 1 CPU for the environment {}".format(args.env)
    kwargs = dict(env_kwargs or {})
    kwargs.setdefault("srl_model", getattr(args, "srl_model", "raw_pixels"))
    # learned SRL models (registered_srl[...][0] is SRLType.SRL): the reference starts a MultiprocessSRLModel server and hands
    # every env a queue pair (:213-216); HipVecEnv loads the encoder once on the GPU (env_kwargs["srl_model_path"],
    # ["state_dim"]) and returns its states as the observation
    envs = HipVecEnv(args.env, args.num_cpu, seed=args.seed, env_kwargs=kwargs, log_dir=getattr(args, "log_dir", None),
                     device_id=getattr(args, "device_id", 0), allow_early_resets=allow_early_resets)
    envs = VecFrameStack(envs, getattr(args, "num_stack", 1))
    if kwargs["srl_model"] != "raw_pixels":
        envs = VecNormalize(envs, norm_obs=True, norm_reward=False)
        if load_path_normalise is not None:
            envs.training = False
            envs.load_running_average(load_path_normalise)
    return envs
