"""environments/mobile_robot/mobile_robot_env.py — constants (:13-28) and class, HIP-backed."""
from srlhip.envs import MobileRobotGymEnv as _Impl

MAX_STEPS = 250
REWARD_DIST_THRESHOLD = 0.4
RENDER_HEIGHT = 224
RENDER_WIDTH = 224
N_DISCRETE_ACTIONS = 4
DELTA_POS = 0.1
RELATIVE_POS = True
NOISE_STD = 0.0
ROBOT_WIDTH = 0.2
ROBOT_LENGTH = 0.325 * 2


def getGlobals():
    return globals()


class MobileRobotGymEnv(_Impl):
    pass
