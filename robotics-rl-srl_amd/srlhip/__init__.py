"""srlhip — Python host side of the MI355X-native vectorised env stepper
(drop-in for the env-step hot path of araffin/robotics-rl-srl)."""
from . import _lib  # noqa: F401
from ._lib import SrlHipError  # noqa: F401
