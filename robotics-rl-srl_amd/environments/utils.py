"""environments/utils.py:36-57 — makeEnv thunks (per-env path, kept for callers that
still want one env object; createEnvs uses the batched HipVecEnv instead)."""
import importlib


def dynamicEnvLoad(env_id):
    from environments.registry import registered_env
    cls = registered_env[env_id][0]
    return importlib.import_module(cls.__module__), cls.__name__, cls.__module__


def makeEnv(env_id, seed, rank, log_dir, allow_early_resets=False, env_kwargs=None):
    def _thunk():
        from environments.registry import registered_env
        local_env_kwargs = dict(env_kwargs or {})
        local_env_kwargs["env_rank"] = rank
        env = registered_env[env_id][0](**local_env_kwargs)
        env.seed(seed + rank)
        return env
    return _thunk
