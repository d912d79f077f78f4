"""environments/kuka_gym/kuka_rand_button_gym_env.py — constants (:3-4) and class, HIP-backed."""
from srlhip.envs import KukaRandButtonGymEnv as _Impl
from .kuka_button_gym_env import *  # noqa: F401,F403
from .kuka_button_gym_env import KukaButtonGymEnv

MAX_STEPS = 1000
BALL_FORCE = 10


def getGlobals():
    return globals()


class KukaRandButtonGymEnv(_Impl, KukaButtonGymEnv):
    pass
