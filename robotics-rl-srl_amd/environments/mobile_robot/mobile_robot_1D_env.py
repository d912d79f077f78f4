from srlhip.envs import MobileRobot1DGymEnv as _Impl
from .mobile_robot_env import *  # noqa: F401,F403
from .mobile_robot_env import MobileRobotGymEnv

N_DISCRETE_ACTIONS = 2


def getGlobals():
    return globals()


class MobileRobot1DGymEnv(_Impl, MobileRobotGymEnv):
    pass
