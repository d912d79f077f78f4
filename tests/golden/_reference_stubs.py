"""Stub modules that let the reference's env *source files* be imported in the
build container, where pybullet / gym / cv2 / srl_zoo are not installed.

Used ONLY by the golden-vector generators in this directory (which run in the
build container, where /root/reference exists).  Nothing here is imported by
the product or at test time: the generators' outputs are committed as .npz.

What is stubbed, and why that does not weaken the pin:
  * ``pybullet``: every ``p.*`` call on the MobileRobot path is rendering or
    scene loading ("no physical effect on pos"), so a no-op
    fake leaves the reference's numpy arithmetic untouched.  For the Kuka
    wrapper a *scripted* fake is used instead (make_kuka_wrapper_golden.py).
  * ``gym``: ``Env`` base class, ``spaces`` containers and
    ``utils.seeding.np_random`` — the latter is the restatement in
    oracle/gym_seeding.py (published algorithm of gym==0.11.0).
  * ``state_representation.episode_saver`` / ``srl_zoo``: recording side-car,
    never instantiated with ``record_data=False``.
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REFERENCE = "/root/reference"


class _AnyCall(types.ModuleType):
    """Module whose unknown attributes are no-op callables returning 0."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)

        def _noop(*args, **kwargs):
            return 0
        return _noop


def make_fake_pybullet(height=224, width=224):
    p = _AnyCall("pybullet")
    p.ER_TINY_RENDERER = 0
    p.DIRECT, p.GUI, p.SHARED_MEMORY = 2, 1, 3
    p.POSITION_CONTROL = 2

    def getCameraImage(width=width, height=height, **kwargs):
        px = np.zeros((height, width, 4), dtype=np.uint8)
        return width, height, px, None, None

    def getQuaternionFromEuler(e):
        return (0.0, 0.0, 0.0, 1.0)

    p.getCameraImage = getCameraImage
    p.getQuaternionFromEuler = getQuaternionFromEuler
    p.computeViewMatrixFromYawPitchRoll = lambda **kw: [0.0] * 16
    p.computeProjectionMatrixFOV = lambda **kw: [0.0] * 16
    return p


def install(pybullet_module=None, real_pybullet_data=False):
    """Put the stubs in sys.modules and the reference on sys.path.  real_pybullet_data: keep the importable pybullet_data
    (make_kuka_pybullet_golden.py runs the reference against the real PyBullet)."""
    sys.path.insert(0, REPO)
    from oracle import gym_seeding

    # --- gym ---------------------------------------------------------------
    gym = types.ModuleType("gym")

    class Env(object):
        metadata = {}
        spec = None

        @property
        def unwrapped(self):
            return self

    class Discrete(object):
        def __init__(self, n):
            self.n = n
            self.np_random = np.random.RandomState()

        def seed(self, seed):
            self.np_random.seed(seed)

        def sample(self):
            return self.np_random.randint(self.n)

    class Box(object):
        def __init__(self, low=None, high=None, shape=None, dtype=None):
            if shape is None:
                self.low, self.high = np.asarray(low), np.asarray(high)
                shape = self.low.shape
            else:
                self.low = np.full(shape, low)
                self.high = np.full(shape, high)
            self.shape, self.dtype = tuple(shape), dtype
            self.np_random = np.random.RandomState()

        def seed(self, seed):
            self.np_random.seed(seed)

        def sample(self):
            return self.np_random.uniform(self.low, self.high, self.shape).astype(self.dtype)

    spaces = types.ModuleType("gym.spaces")
    spaces.Discrete, spaces.Box = Discrete, Box
    utils = types.ModuleType("gym.utils")
    seeding = types.ModuleType("gym.utils.seeding")
    seeding.np_random = gym_seeding.np_random
    utils.seeding = seeding
    gym.Env, gym.spaces, gym.utils = Env, spaces, utils
    sys.modules.update({"gym": gym, "gym.spaces": spaces, "gym.utils": utils,
                        "gym.utils.seeding": seeding})

    # --- pybullet (+ data) -------------------------------------------------
    p = pybullet_module if pybullet_module is not None else make_fake_pybullet()
    sys.modules["pybullet"] = p
    if not real_pybullet_data:
        pd = types.ModuleType("pybullet_data")
        pd.getDataPath = lambda: "/nonexistent/pybullet_data"
        sys.modules["pybullet_data"] = pd

    # --- recording side-car / srl_zoo ---------------------------------------
    sr = types.ModuleType("state_representation")
    sr.__path__ = []
    es = types.ModuleType("state_representation.episode_saver")

    class EpisodeSaver(object):
        def __init__(self, *a, **k):
            raise RuntimeError("EpisodeSaver is stubbed: record_data must be False")
    es.EpisodeSaver = EpisodeSaver
    sr.episode_saver = es
    sys.modules["state_representation"] = sr
    sys.modules["state_representation.episode_saver"] = es
    zoo = types.ModuleType("srl_zoo")
    zoo.__path__ = []
    pre = types.ModuleType("srl_zoo.preprocessing")
    pre.getNChannels = lambda: 3
    zoo.preprocessing = pre
    sys.modules["srl_zoo"] = zoo
    sys.modules["srl_zoo.preprocessing"] = pre

    # reference package root (environments/…); must precede the repo's mirror
    if REFERENCE not in sys.path:
        sys.path.insert(0, REFERENCE)
    return p
