"""environments/kuka_gym/kuka_2button_gym_env.py — constants (:3) and class, HIP-backed."""
from srlhip.envs import Kuka2ButtonGymEnv as _Impl
from .kuka_button_gym_env import *  # noqa: F401,F403
from .kuka_button_gym_env import KukaButtonGymEnv

MAX_STEPS = 1500


def getGlobals():
    return globals()


class Kuka2ButtonGymEnv(_Impl, KukaButtonGymEnv):
    pass
