"""environments/kuka_gym/kuka_button_gym_env.py — constants (:17-35) and class, HIP-backed."""
from srlhip.envs import KukaButtonGymEnv as _Impl

MAX_STEPS = 1000
N_CONTACTS_BEFORE_TERMINATION = 5
N_STEPS_OUTSIDE_SAFETY_SPHERE = 5000
RENDER_HEIGHT = 224
RENDER_WIDTH = 224
Z_TABLE = -0.2
N_DISCRETE_ACTIONS = 6
BUTTON_LINK_IDX = 1
BUTTON_GLIDER_IDX = 1
DELTA_V = 0.03
DELTA_V_CONTINUOUS = 0.0035
DELTA_THETA = 0.1
RELATIVE_POS = True
NOISE_STD = 0.01
NOISE_STD_CONTINUOUS = 0.0001
NOISE_STD_JOINTS = 0.002
N_RANDOM_ACTIONS_AT_INIT = 5
BUTTON_DISTANCE_HEIGHT = 0.28


def getGlobals():
    return globals()


class KukaButtonGymEnv(_Impl):
    pass
