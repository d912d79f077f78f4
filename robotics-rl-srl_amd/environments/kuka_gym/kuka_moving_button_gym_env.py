"""environments/kuka_gym/kuka_moving_button_gym_env.py — constants (:3-6) and class, HIP-backed."""
from srlhip.envs import KukaMovingButtonGymEnv as _Impl
from .kuka_button_gym_env import *  # noqa: F401,F403
from .kuka_button_gym_env import KukaButtonGymEnv

MAX_STEPS = 1500
BUTTON_SPEED = 0.001
BUTTON_YMIN = -0.3
BUTTON_YMAX = 0.3


def getGlobals():
    return globals()


class KukaMovingButtonGymEnv(_Impl, KukaButtonGymEnv):
    pass
