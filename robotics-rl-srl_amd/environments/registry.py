"""environments/registry.py:41-53 — name -> (class, super class, plot type, threading type)
for the envs libsrlhip steps: every simulated Kuka / MobileRobot env of the reference.  The real-robot
(Baxter, Robobo, Omnirobot) and CarRacing entries are not part of this hot path."""
from environments import PlottingType, ThreadingType
from environments.srl_env import SRLGymEnv
from environments.kuka_gym.kuka_button_gym_env import KukaButtonGymEnv
from environments.kuka_gym.kuka_moving_button_gym_env import KukaMovingButtonGymEnv
from environments.kuka_gym.kuka_2button_gym_env import Kuka2ButtonGymEnv
from environments.kuka_gym.kuka_rand_button_gym_env import KukaRandButtonGymEnv
from environments.mobile_robot.mobile_robot_env import MobileRobotGymEnv
from environments.mobile_robot.mobile_robot_2target_env import MobileRobot2TargetGymEnv
from environments.mobile_robot.mobile_robot_1D_env import MobileRobot1DGymEnv
from environments.mobile_robot.mobile_robot_line_target_env import MobileRobotLineTargetGymEnv

registered_env = {
    "KukaButtonGymEnv-v0":            (KukaButtonGymEnv, SRLGymEnv, PlottingType.PLOT_3D, ThreadingType.PROCESS),
    "KukaMovingButtonGymEnv-v0":      (KukaMovingButtonGymEnv, KukaButtonGymEnv, PlottingType.PLOT_3D, ThreadingType.PROCESS),
    "KukaRandButtonGymEnv-v0":        (KukaRandButtonGymEnv, KukaButtonGymEnv, PlottingType.PLOT_3D, ThreadingType.PROCESS),
    "Kuka2ButtonGymEnv-v0":           (Kuka2ButtonGymEnv, KukaButtonGymEnv, PlottingType.PLOT_3D, ThreadingType.PROCESS),
    "MobileRobotGymEnv-v0":           (MobileRobotGymEnv, SRLGymEnv, PlottingType.PLOT_2D, ThreadingType.PROCESS),
    "MobileRobot2TargetGymEnv-v0":    (MobileRobot2TargetGymEnv, MobileRobotGymEnv, PlottingType.PLOT_2D, ThreadingType.PROCESS),
    "MobileRobot1DGymEnv-v0":         (MobileRobot1DGymEnv, MobileRobotGymEnv, PlottingType.PLOT_2D, ThreadingType.PROCESS),
    "MobileRobotLineTargetGymEnv-v0": (MobileRobotLineTargetGymEnv, MobileRobotGymEnv, PlottingType.PLOT_2D, ThreadingType.PROCESS),
}
