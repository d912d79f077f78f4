"""Per-env gym surface (environments/srl_env.py:5-102) on top of libsrlhip.

Each class keeps the reference's constructor kwargs (unknown ones are swallowed
with ``**_``), spaces, ``seed/reset/step/render/close`` and the SRL getters, but
owns a one-env handle on the GPU instead of a PyBullet client.  Use
``srlhip.vec_env.HipVecEnv`` for throughput: a single-env handle pays a kernel
launch per step and is only meant to keep per-env call sites working
(``dataset_generator``, debugging, ``enjoy``)."""
import numpy as np

from . import _lib
from .gym_compat import Box, Discrete, Env, np_random

OBS_MODES = {"ground_truth": _lib.OBS_GROUND_TRUTH, "joints": _lib.OBS_JOINTS,
             "joints_position": _lib.OBS_JOINTS_POSITION, "raw_pixels": _lib.OBS_RAW_PIXELS}


class SRLGymEnv(Env):
    """environments/srl_env.py:SRLGymEnv"""
    metadata = {'render.modes': ['human', 'rgb_array'], 'video.frames_per_second': 50}

    def __init__(self, *, srl_model, relative_pos, env_rank, srl_pipe):
        self.env_rank = env_rank
        self.srl_pipe = srl_pipe
        self.srl_model = srl_model
        self.relative_pos = relative_pos
        self.np_random = None
        self.seed(0)

    def getSRLState(self, observation):
        if self.srl_model == "ground_truth":
            if self.relative_pos:
                return self.getGroundTruth() - self.getTargetPos()
            return self.getGroundTruth()
        self.srl_pipe[0].put((self.env_rank, observation))
        return self.srl_pipe[1][self.env_rank].get()

    def getTargetPos(self):
        raise NotImplementedError()

    @staticmethod
    def getGroundTruthDim():
        raise NotImplementedError()

    def getGroundTruth(self):
        raise NotImplementedError()

    def seed(self, seed=None):
        self.np_random, seed = np_random(seed)
        return [seed]

    def close(self):
        pass

    def step(self, action):
        raise NotImplementedError()

    def reset(self):
        raise NotImplementedError()

    def render(self, mode='human'):
        raise NotImplementedError()


class _HipEnv(SRLGymEnv):
    """Shared machinery: a one-env handle with device-resident np_random (MT19937)."""
    ENV_KIND = None
    RELATIVE_POS = True

    def _open(self, *, is_discrete, random_target, shape_reward, srl_model, max_distance, force_down=True,
              action_repeat=1, action_joints=False, multi_view=False, device_id=0, img_hw=(224, 224)):
        cfg = _lib.default_config(self.ENV_KIND)
        cfg.num_envs, cfg.device_id = 1, device_id
        cfg.is_discrete, cfg.random_target, cfg.shape_reward = int(is_discrete), int(random_target), int(shape_reward)
        cfg.force_down, cfg.action_repeat, cfg.action_joints = int(force_down), int(action_repeat), int(action_joints)
        cfg.multi_view, cfg.max_distance = int(multi_view), float(max_distance)
        cfg.img_h, cfg.img_w = img_hw
        self._obs_is_image = srl_model == "raw_pixels"
        # the stepper itself always produces the ground-truth state; images come from render()
        cfg.obs_mode = OBS_MODES.get(srl_model, _lib.OBS_GROUND_TRUTH) if not self._obs_is_image else _lib.OBS_GROUND_TRUTH
        cfg.rng_mode, cfg.auto_reset, cfg.io_device = _lib.RNG_MT19937, 0, 0
        self._h = _lib.Handle(cfg)                      # raises without libsrlhip.so / a GPU
        self._seeded = False

    def seed(self, seed=None):
        out = super(_HipEnv, self).seed(seed)
        if getattr(self, "_h", None) is not None:
            self._h.seed(np.array([out[0] % 2 ** 63], dtype=np.int64))
        return out

    def close(self):
        if getattr(self, "_h", None) is not None:
            self._h.close()
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:       # noqa: BLE001
            pass

    def _f(self, field):
        return self._h.get_state(field)

    @classmethod
    def _globals(cls):
        """getGlobals() of the module that defines the env class (train.py:291-307, EpisodeSaver)."""
        import importlib
        mod = importlib.import_module(cls.__module__)
        return mod.getGlobals() if hasattr(mod, "getGlobals") else {}

    def _reward_value(self):
        r = float(self._f(_lib.F_LAST_REWARD)[0])
        if not self._shape_reward:
            return int(r)                  # the reference returns a Python int for sparse rewards
        return r

    def _obs(self):
        if self._obs_is_image:
            return self.render("rgb_array")
        return self.getSRLState(None)

    def render(self, mode='human', close=False):
        if mode != "rgb_array":
            return np.array([])
        return self._h.render()[0]          # HIP tile rasteriser, uint8 [H][W][3 (6 with multi_view)]


# ---------------------------------------------------------------------------- MobileRobot
class MobileRobotGymEnv(_HipEnv):
    """environments/mobile_robot/mobile_robot_env.py:MobileRobotGymEnv"""
    ENV_KIND = _lib.ENV_MOBILE
    N_DISCRETE_ACTIONS = 4

    def __init__(self, urdf_root=None, renders=False, is_discrete=True, name="mobile_robot", max_distance=1.6,
                 shape_reward=False, record_data=False, srl_model="raw_pixels", random_target=False, force_down=True,
                 state_dim=-1, learn_states=False, verbose=False, save_path='srl_zoo/data/', env_rank=0,
                 srl_pipe=None, fpv=False, device_id=0, **_):
        self._h = None
        super(MobileRobotGymEnv, self).__init__(srl_model=srl_model, relative_pos=self.RELATIVE_POS,
                                                env_rank=env_rank, srl_pipe=srl_pipe)
        self._is_discrete, self._random_target, self._shape_reward = is_discrete, random_target, shape_reward
        self._max_distance, self._force_down, self.fpv, self.verbose = max_distance, force_down, fpv, verbose
        self._width = self._height = 224
        self.max_steps = 250
        self.state_dim = state_dim
        self.terminated = False
        self.has_bumped = False
        self.saver = None
        if record_data:
            from .recorder import EpisodeSaver
            self.saver = EpisodeSaver(name, max_distance, state_dim, globals_=self._globals(), relative_pos=True,
                                      learn_states=learn_states, path=save_path)
        if not is_discrete and self.ENV_KIND in (_lib.ENV_MOBILE_1D, _lib.ENV_MOBILE_2TARGET):
            raise ValueError("Only discrete actions is supported")
        self._open(is_discrete=is_discrete, random_target=random_target, shape_reward=shape_reward,
                   srl_model=srl_model, max_distance=max_distance, device_id=device_id, multi_view=bool(fpv))
        self.seed(0)
        if is_discrete:
            self.action_space = Discrete(self.N_DISCRETE_ACTIONS)
        else:
            self.action_space = Box(low=-1, high=1, shape=(2,), dtype=np.float32)
        if self.srl_model == "ground_truth":
            self.state_dim = self.getGroundTruthDim()
        if self.srl_model == "raw_pixels":
            # fpv=True stacks the car-camera view behind the top-down one (mobile_robot_env.py:313-332, getNChannels)
            self.observation_space = Box(low=0, high=255, shape=(self._height, self._width, 6 if fpv else 3), dtype=np.uint8)
        else:
            self.observation_space = Box(low=-np.inf, high=np.inf, shape=(self.state_dim,), dtype=np.float32)

    @property
    def robot_pos(self):
        return np.array([self._f(_lib.F_POS_X)[0], self._f(_lib.F_POS_Y)[0], 0.0])

    @property
    def target_pos(self):
        if int(self._f(_lib.F_CUR_TARGET)[0]):
            return np.array([self._f(_lib.F_TARGET2_X)[0], self._f(_lib.F_TARGET2_Y)[0], 0.0])
        return np.array([self._f(_lib.F_TARGET_X)[0], self._f(_lib.F_TARGET_Y)[0], 0.0])

    @property
    def _env_step_counter(self):
        return int(self._f(_lib.F_STEP_COUNT)[0])

    def getTargetPos(self):
        return self.target_pos[:2]

    @staticmethod
    def getGroundTruthDim():
        return 2

    def getGroundTruth(self):
        return np.array(self.robot_pos)[:2]

    def reset(self):
        self.terminated = False
        self._h.reset()
        obs = self._obs()
        if self.saver is not None:          # the recorder always stores the rendered frame, whatever srl_model returns (kuka_button_gym_env.py:276-277)
            self.saver.reset(self.render("rgb_array"), self.getTargetPos(), self.getGroundTruth())
        return np.array(obs)

    def step(self, action):
        if self._is_discrete:
            a = np.array([int(action)], dtype=np.int32)
        else:
            a = np.asarray(action, dtype=np.float32).reshape(1, 2)
        _, _, done = self._h.step(a)
        reward = self._reward_value()
        self.has_bumped = (reward == -1) if not self._shape_reward else False
        obs = self._obs()
        done = bool(done[0])
        if self.saver is not None:
            self.saver.step(self.render("rgb_array"), action, reward, done, self.getGroundTruth())
        return np.array(obs), reward, done, {}


class MobileRobot1DGymEnv(MobileRobotGymEnv):
    ENV_KIND = _lib.ENV_MOBILE_1D
    N_DISCRETE_ACTIONS = 2

    def __init__(self, name="mobile_robot_1D", **kwargs):
        super(MobileRobot1DGymEnv, self).__init__(name=name, **kwargs)

    def getTargetPos(self):
        return self.target_pos[:1]

    @staticmethod
    def getGroundTruthDim():
        return 1

    def getGroundTruth(self):
        return np.array(self.robot_pos)[:1]


class MobileRobot2TargetGymEnv(MobileRobotGymEnv):
    ENV_KIND = _lib.ENV_MOBILE_2TARGET

    def __init__(self, name="mobile_robot_2target", **kwargs):
        super(MobileRobot2TargetGymEnv, self).__init__(name=name, **kwargs)

    @property
    def current_target(self):
        return int(self._f(_lib.F_CUR_TARGET)[0])


class MobileRobotLineTargetGymEnv(MobileRobotGymEnv):
    ENV_KIND = _lib.ENV_MOBILE_LINE
    ROBOT_OFFSET = 0.2

    def __init__(self, name="mobile_robot_line_target", **kwargs):
        super(MobileRobotLineTargetGymEnv, self).__init__(name=name, **kwargs)

    def getTargetPos(self):
        return self.target_pos[:1] - self.ROBOT_OFFSET


# ---------------------------------------------------------------------------- Kuka
class _KukaProxy(object):
    """Stand-in for environments/kuka_gym/kuka.py:Kuka attributes used by callers."""
    joint_positions = [0.006418, 0.113184, -0.011401, -1.289317, 0.005379, 1.737684, -0.006539, 0.000048,
                       -0.299912, 0.000000, -0.000043, 0.299960, 0.000000, -0.000200]
    kuka_end_effector_index = 6
    kuka_gripper_index = 8
    kuka_uid = 0

    def __init__(self, env):
        self._env = env

    @property
    def end_effector_pos(self):
        return self._env._f(_lib.F_KUKA_EE_TARGET)[:, 0].copy()


class KukaButtonGymEnv(_HipEnv):
    """environments/kuka_gym/kuka_button_gym_env.py:KukaButtonGymEnv"""
    ENV_KIND = _lib.ENV_KUKA_BUTTON

    def __init__(self, urdf_root=None, renders=False, is_discrete=True, multi_view=False, name="kuka_button_gym",
                 max_distance=0.8, action_repeat=1, shape_reward=False, action_joints=False, record_data=False,
                 random_target=False, force_down=True, state_dim=-1, learn_states=False, verbose=False,
                 save_path='srl_zoo/data/', env_rank=0, srl_pipe=None, srl_model="raw_pixels", device_id=0, **_):
        self._h = None
        super(KukaButtonGymEnv, self).__init__(srl_model=srl_model, relative_pos=self.RELATIVE_POS, env_rank=env_rank,
                                               srl_pipe=srl_pipe)
        self._is_discrete, self._random_target, self._shape_reward = is_discrete, random_target, shape_reward
        self._max_distance, self._force_down, self._action_repeat = max_distance, force_down, action_repeat
        self.action_joints, self.multi_view, self.verbose = action_joints, multi_view, verbose
        self._width = self._height = 224
        self.max_steps = 1000
        self.state_dim = state_dim
        self.action = None
        self.saver = None
        if record_data:
            from .recorder import EpisodeSaver
            self.saver = EpisodeSaver(name, max_distance, state_dim, globals_=self._globals(), relative_pos=True, learn_states=learn_states, path=save_path)
        self._open(is_discrete=is_discrete, random_target=random_target, shape_reward=shape_reward,
                   srl_model=srl_model, max_distance=max_distance, force_down=force_down, action_repeat=action_repeat,
                   action_joints=action_joints, multi_view=multi_view, device_id=device_id)
        self._kuka = _KukaProxy(self)
        self.seed(0)
        if is_discrete:
            self.action_space = Discrete(6)
        else:
            action_dim = 7 if action_joints else 3
            self._action_bound = 1
            high = np.array([self._action_bound] * action_dim)
            self.action_space = Box(-high, high, dtype=np.float32)
        if srl_model == "ground_truth":
            self.state_dim = self.getGroundTruthDim()
        elif srl_model == "joints":
            self.state_dim = self.getJointsDim()
        elif srl_model == "joints_position":
            self.state_dim = self.getGroundTruthDim() + self.getJointsDim()
        if srl_model == "raw_pixels":
            self.observation_space = Box(low=0, high=255, shape=(self._height, self._width, 3), dtype=np.uint8)
        else:
            self.observation_space = Box(low=-np.inf, high=np.inf, shape=(self.state_dim,), dtype=np.float32)

    # counters mirrored from the device
    @property
    def _env_step_counter(self):
        return int(self._f(_lib.F_STEP_COUNT)[0])

    @property
    def n_contacts(self):
        return int(self._f(_lib.F_KUKA_COUNTERS)[0, 0])

    @property
    def n_steps_outside(self):
        return int(self._f(_lib.F_KUKA_COUNTERS)[1, 0])

    @property
    def terminated(self):
        return bool(self._f(_lib.F_KUKA_COUNTERS)[2, 0])

    @property
    def button_pos(self):
        return self._f(_lib.F_KUKA_BUTTON_POS)[:, 0].copy()

    def getSRLState(self, observation):
        state = []
        if self.srl_model in ["ground_truth", "joints_position"]:
            if self.relative_pos:
                state += list(self.getGroundTruth() - self.getTargetPos())
            else:
                state += list(self.getGroundTruth())
        if self.srl_model in ["joints", "joints_position"]:
            state += list(self._kuka.joint_positions)
        if len(state) != 0:
            return np.array(state)
        self.srl_pipe[0].put((self.env_rank, observation))
        return self.srl_pipe[1][self.env_rank].get()

    def getTargetPos(self):
        return self.button_pos

    @staticmethod
    def getJointsDim():
        return 14

    @staticmethod
    def getGroundTruthDim():
        return 3

    def getGroundTruth(self):
        return np.array(self.getArmPos())

    def getArmPos(self):
        return tuple(self._f(_lib.F_KUKA_GRIPPER)[:, 0])

    def reset(self):
        self._h.reset()
        obs = self._obs()
        if self.saver is not None:          # the recorder always stores the rendered frame, whatever srl_model returns (kuka_button_gym_env.py:276-277)
            self.saver.reset(self.render("rgb_array"), self.getTargetPos(), self.getGroundTruth())
        return np.array(obs)

    def step(self, action):
        self.action = action if action is not None else self.action
        if self._is_discrete:
            a = np.array([-1 if action is None else int(action)], dtype=np.int32)
        else:
            dim = 7 if self.action_joints else 3
            if action is None:
                raise NotImplementedError("None action with continuous actions: use the discrete env or HipVecEnv")
            a = np.asarray(action, dtype=np.float32).reshape(1, dim)
        _, _, done = self._h.step(a)
        reward = self._reward_value()
        obs = self._obs()
        done = bool(done[0])
        if self.saver is not None:
            self.saver.step(self.render("rgb_array"), self.action, reward, done, self.getGroundTruth())
        return np.array(obs), reward, done, {}


class KukaMovingButtonGymEnv(KukaButtonGymEnv):
    """environments/kuka_gym/kuka_moving_button_gym_env.py:KukaMovingButtonGymEnv"""
    ENV_KIND = _lib.ENV_KUKA_MOVING

    def __init__(self, name="kuka_moving_button_gym", **kwargs):
        super(KukaMovingButtonGymEnv, self).__init__(name=name, **kwargs)
        self.max_steps = 1500


class Kuka2ButtonGymEnv(KukaButtonGymEnv):
    """environments/kuka_gym/kuka_2button_gym_env.py:Kuka2ButtonGymEnv — two buttons pressed in order (y = +0.125 first,
    then the darker one at y = -0.125); ctor defaults max_distance=2, force_down=False (:29), max_steps 1500 (:3, :33)."""
    ENV_KIND = _lib.ENV_KUKA_2BUTTON

    def __init__(self, name="kuka_2button_gym", max_distance=2, force_down=False, **kwargs):
        super(Kuka2ButtonGymEnv, self).__init__(name=name, max_distance=max_distance, force_down=force_down, **kwargs)
        self.max_steps = 1500

    @property
    def n_contacts(self):
        return [int(self._f(_lib.F_KUKA_COUNTERS)[0, 0]), int(self._f(_lib.F_KUKA_GOAL)[1, 0])]

    @property
    def goal_id(self):
        return int(self._f(_lib.F_KUKA_GOAL)[0, 0])

    @property
    def button_all_pos(self):
        z = self.button_pos[2]
        b1, b2 = self._f(_lib.F_KUKA_BUTTON_XY)[:, 0], self._f(_lib.F_KUKA_BUTTON2_XY)[:, 0]
        return [np.array([b1[0], b1[1], z]), np.array([b2[0], b2[1], z])]


class KukaRandButtonGymEnv(KukaButtonGymEnv):
    """environments/kuka_gym/kuka_rand_button_gym_env.py:KukaRandButtonGymEnv — KukaButtonGymEnv plus ten randomly placed
    distractor objects and a ball.  reset() consumes the reference's 20 extra np_random draws and applies its keep rule;
    the objects are scenery for the rasteriser (their rigid-body dynamics, their types and the push on the ball come from
    pybullet / the global unseeded np.random in the reference and are not modelled)."""
    ENV_KIND = _lib.ENV_KUKA_RAND

    def __init__(self, name="kuka_rand_button_gym", **kwargs):
        super(KukaRandButtonGymEnv, self).__init__(name=name, **kwargs)
        self.max_steps = 1000

    @property
    def objects(self):
        """(x, y, present) of the ten candidate distractors of the current episode"""
        return self._f(_lib.F_KUKA_OBJECTS)[:, 0].reshape(10, 3).copy()


ENV_CLASSES = {
    "KukaButtonGymEnv-v0": KukaButtonGymEnv,
    "KukaRandButtonGymEnv-v0": KukaRandButtonGymEnv,
    "KukaMovingButtonGymEnv-v0": KukaMovingButtonGymEnv,
    "Kuka2ButtonGymEnv-v0": Kuka2ButtonGymEnv,
    "MobileRobotGymEnv-v0": MobileRobotGymEnv,
    "MobileRobot2TargetGymEnv-v0": MobileRobot2TargetGymEnv,
    "MobileRobot1DGymEnv-v0": MobileRobot1DGymEnv,
    "MobileRobotLineTargetGymEnv-v0": MobileRobotLineTargetGymEnv,
}
