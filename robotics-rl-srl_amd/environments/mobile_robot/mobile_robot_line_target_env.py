from srlhip.envs import MobileRobotLineTargetGymEnv as _Impl
from .mobile_robot_env import *  # noqa: F401,F403
from .mobile_robot_env import MobileRobotGymEnv

REWARD_DIST_THRESHOLD = 0.1
ROBOT_OFFSET = 0.2


def getGlobals():
    return globals()


class MobileRobotLineTargetGymEnv(_Impl, MobileRobotGymEnv):
    pass
