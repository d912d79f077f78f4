from srlhip.envs import MobileRobot2TargetGymEnv as _Impl
from .mobile_robot_env import *  # noqa: F401,F403
from .mobile_robot_env import MobileRobotGymEnv

MAX_STEPS = 1500     # dead in the reference too: the base ctor reads its own module's 250


def getGlobals():
    return globals()


class MobileRobot2TargetGymEnv(_Impl, MobileRobotGymEnv):
    pass
