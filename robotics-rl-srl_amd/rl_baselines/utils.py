"""rl_baselines/utils.py:194-229 — createEnvs with the batched GPU vec env spliced in
where the reference builds `[makeEnv(...)] -> DummyVecEnv | SubprocVecEnv` (:213-220).
Everything after that line is unchanged: VecFrameStack, then VecNormalize for non-pixel
observations.

Multi-GPU: rl_baselines/train.py is ONE process, so the node is used from inside this VecEnv — `args.device_ids` ("all", "0,1,2,3" or
a list; `--device-ids` on the command lines of this package) shards the `num_cpu` envs over those GPUs (contiguous blocks of global
env ids, env i seeded seed + i as at environments/utils.py:52 whatever the number of GPUs); without it the env runs on
`args.device_id` (default 0)."""
from environments import ThreadingType

try:                                        # pragma: no cover - stable_baselines is absent here
    from stable_baselines.common.vec_env import VecNormalize, VecFrameStack
except Exception:                           # noqa: BLE001
    from srlhip.vec_wrappers import VecNormalize, VecFrameStack
from srlhip.vec_env import HipVecEnv, parse_device_ids


def createEnvs(args, allow_early_resets=False, env_kwargs=None, load_path_normalise=None):
    from environments.registry import registered_env
    assert not (registered_env[args.env][3] is ThreadingType.NONE and args.num_cpu != 1), \
        "Error: cannot have more than 1 CPU for the environment {}".format(args.env)
    kwargs = dict(env_kwargs or {})
    kwargs.setdefault("srl_model", getattr(args, "srl_model", "raw_pixels"))
    # learned SRL models (registered_srl[...][0] is SRLType.SRL): the reference starts a MultiprocessSRLModel server and hands
    # every env a queue pair (:213-216); HipVecEnv loads the encoder once on the GPU (env_kwargs["srl_model_path"],
    # ["state_dim"]) and returns its states as the observation
    device_ids = parse_device_ids(getattr(args, "device_ids", None))
    if device_ids is not None:
        device_ids = device_ids[:max(1, min(len(device_ids), args.num_cpu))]          # never more shards than envs
    envs = HipVecEnv(args.env, args.num_cpu, seed=args.seed, env_kwargs=kwargs, log_dir=getattr(args, "log_dir", None),
                     device_id=getattr(args, "device_id", 0), device_ids=device_ids, allow_early_resets=allow_early_resets,
                     persistent=True if getattr(args, "persistent", False) else None)      # (--persistent: no kernel launch per step, srlhip_set_persistent)
    envs = VecFrameStack(envs, getattr(args, "num_stack", 1))
    if kwargs["srl_model"] != "raw_pixels":
        envs = VecNormalize(envs, norm_obs=True, norm_reward=False)
        if load_path_normalise is not None:
            envs.training = False
            envs.load_running_average(load_path_normalise)
    return envs
