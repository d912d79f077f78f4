"""Drop-in mirror of the reference's `environments` package for the env-step hot
path: same module paths, class names, `registered_env` table and `getGlobals()`,
backed by libsrlhip (MI355X) instead of PyBullet."""
from enum import Enum


class PlottingType(Enum):
    PLOT_2D = 1
    PLOT_3D = 2


class ThreadingType(Enum):
    PROCESS = 1
    THREADING = 2
    NONE = 3
