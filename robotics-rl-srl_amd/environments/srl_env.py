from srlhip.envs import SRLGymEnv  # noqa: F401  (environments/srl_env.py:5-102)
