"""GPU parity of the tile rasteriser (raw_pixels observation) against oracle/raster_oracle.c:
uint8 images bit-exact (float32, no contraction on either side)."""
import numpy as np
import pytest

from oracle import raster_clib
from srlhip import _lib

pytestmark = pytest.mark.gpu


def kuka_state(h):
    q = h.get_state(_lib.F_KUKA_Q).T
    b = h.get_state(_lib.F_KUKA_BUTTON_Q)[0]
    bp = h.get_state(_lib.F_KUKA_BUTTON_POS).T
    return np.concatenate([q, b[:, None], bp[:, :2]], axis=1)


def gq(h):
    """q of the gripper joints (full-model handles: the rasteriser draws the fingers from them)"""
    return h.get_state(_lib.F_KUKA_GRIPPER_Q).T


def mobile_state(h):
    f = lambda k: h.get_state(k)
    return np.stack([f(_lib.F_POS_X), f(_lib.F_POS_Y), f(_lib.F_TARGET_X), f(_lib.F_TARGET_Y), f(_lib.F_TARGET2_X), f(_lib.F_TARGET2_Y)], 1)


def assert_images_equal(gpu, ora):
    assert gpu.shape == ora.shape and gpu.dtype == np.uint8
    diff = np.abs(gpu.astype(np.int16) - ora.astype(np.int16))
    frac = (diff.max(axis=-1) > 0).mean()
    assert frac == 0.0, "differing pixels: {:.5f}, max diff {}".format(frac, diff.max())


# (50, 38): width not a multiple of 4 (byte stores); (16, 1024): the widest frame, two-row bands; (300, 12): tall and narrow, two cameras
@pytest.mark.parametrize("hw,multi_view", [((64, 64), 0), ((224, 224), 0), ((64, 64), 1), ((48, 80), 0), ((50, 38), 0), ((16, 1024), 0), ((300, 12), 1)])
def test_kuka_images_match_oracle(hw, multi_view):
    n = 64
    cfg = _lib.default_config(_lib.ENV_KUKA_BUTTON)
    cfg.num_envs, cfg.seed0, cfg.random_target, cfg.multi_view = n, 2, 1, multi_view
    cfg.obs_mode, cfg.img_h, cfg.img_w = _lib.OBS_RAW_PIXELS, hw[0], hw[1]
    h = _lib.Handle(cfg)
    obs = h.reset()
    assert obs.shape == (n, hw[0], hw[1], 6 if multi_view else 3) and obs.dtype == np.uint8
    assert_images_equal(obs, raster_clib.render(4, kuka_state(h), hw[0], hw[1], multi_view, gripper_q=gq(h)))
    actions = np.random.RandomState(0).randint(6, size=(40, n)).astype(np.int32)
    for t in range(40):
        obs, r, d = h.step(actions[t])
    assert_images_equal(obs, raster_clib.render(4, kuka_state(h), hw[0], hw[1], multi_view, gripper_q=gq(h)))
    assert_images_equal(h.render(), obs)
    assert len(np.unique(obs.reshape(-1, obs.shape[-1])[:, :3], axis=0)) > 50      # a real shaded scene, not a flat fill
    h.close()


@pytest.mark.parametrize("kind", [0, 1, 2, 3])
def test_mobile_images_match_oracle(kind):
    n = 32
    cfg = _lib.default_config(kind)
    cfg.num_envs, cfg.seed0, cfg.random_target = n, 4, 1
    cfg.obs_mode, cfg.img_h, cfg.img_w = _lib.OBS_RAW_PIXELS, 64, 64
    h = _lib.Handle(cfg)
    obs = h.reset()
    assert_images_equal(obs, raster_clib.render(kind, mobile_state(h)))
    out = h.rollout(30, actions=np.random.RandomState(1).randint(2, size=(30, n)).astype(np.int32))
    assert out["obs"].shape == (30, n, 64, 64, 3)
    assert_images_equal(out["obs"][-1], raster_clib.render(kind, mobile_state(h)))
    # the robot (blue box) moves between frames
    assert (out["obs"][0] != out["obs"][-1]).any()
    h.close()


@pytest.mark.parametrize("kind", [0, 1, 2])
def test_mobile_fpv_camera_matches_oracle(kind):
    """fpv=True (mobile_robot_env.py:313-332): the car camera's view is stacked behind the top-down one and follows the robot."""
    n = 32
    cfg = _lib.default_config(kind)
    cfg.num_envs, cfg.seed0, cfg.random_target, cfg.multi_view = n, 5, 1, 1
    cfg.obs_mode, cfg.img_h, cfg.img_w = _lib.OBS_RAW_PIXELS, 64, 64
    h = _lib.Handle(cfg)
    obs = h.reset()
    assert obs.shape == (n, 64, 64, 6)
    assert_images_equal(obs, raster_clib.render(kind, mobile_state(h), multi_view=True))
    plain = raster_clib.render(kind, mobile_state(h))
    assert_images_equal(obs[..., :3], plain)                       # the first three channels are the usual top-down view
    first = obs.copy()
    out = h.rollout(20, actions=np.random.RandomState(2).randint(2, size=(20, n)).astype(np.int32))
    assert_images_equal(out["obs"][-1], raster_clib.render(kind, mobile_state(h), multi_view=True))
    assert (out["obs"][-1][..., 3:] != first[..., 3:]).reshape(n, -1).any(1).mean() > 0.5   # the view moves with the robot
    assert len(np.unique(obs[..., 3:].reshape(-1, 3), axis=0)) > 3   # robot hood, floor checker, wall, sky: not a flat fill
    h.close()


def test_facade_render_and_raw_pixel_vec_env():
    from environments.kuka_gym.kuka_button_gym_env import KukaButtonGymEnv
    from srlhip.vec_env import HipVecEnv
    env = KukaButtonGymEnv()                      # default srl_model = raw_pixels, 224 x 224 like the reference
    assert env.observation_space.shape == (224, 224, 3)
    env.seed(0)
    obs = env.reset()
    assert obs.shape == (224, 224, 3) and obs.dtype == np.uint8 and env.render().size == 0
    obs2, r, d, _ = env.step(4)
    assert obs2.shape == (224, 224, 3)
    env.close()
    from environments.mobile_robot.mobile_robot_env import MobileRobotGymEnv
    fenv = MobileRobotGymEnv(fpv=True)
    assert fenv.observation_space.shape == (224, 224, 6) and fenv.reset().shape == (224, 224, 6)
    fenv.close()
    venv = HipVecEnv("MobileRobotGymEnv-v0", 16, env_kwargs={"srl_model": "raw_pixels", "img_shape": (64, 64)})
    o = venv.reset()
    assert o.shape == (16, 64, 64, 3) and o.dtype == np.uint8
    o, r, d, info = venv.step([0] * 16)
    assert len(venv.get_images()) == 16
    venv.close()


def test_two_button_images_match_oracle():
    """Kuka2ButtonGymEnv scene: the second button (simple_button_2.urdf colours) is rendered at its drawn position."""
    n = 32
    cfg = _lib.default_config(_lib.ENV_KUKA_2BUTTON)
    cfg.num_envs, cfg.seed0, cfg.random_target = n, 4, 1
    cfg.obs_mode, cfg.img_h, cfg.img_w = _lib.OBS_RAW_PIXELS, 64, 64
    h = _lib.Handle(cfg)
    obs = h.reset()

    def state():
        s = kuka_state(h)
        return np.concatenate([s, h.get_state(_lib.F_KUKA_BUTTON2_Q)[0][:, None], h.get_state(_lib.F_KUKA_BUTTON2_XY).T], axis=1)
    assert_images_equal(obs, raster_clib.render(6, state(), 64, 64, gripper_q=gq(h)))
    dark_green = (obs[..., 0] < 80) & (obs[..., 1] > 100) & (obs[..., 2] > 50) & (obs[..., 2] < 130)
    assert dark_green.reshape(n, -1).sum(1).min() > 20          # the darker second cap is visible in every env
    for t in range(30):
        obs, r, d = h.step(np.full(n, 4, np.int32))
    assert_images_equal(obs, raster_clib.render(6, state(), 64, 64, gripper_q=gq(h)))
    h.close()


def test_rand_button_images_match_oracle():
    """KukaRandButtonGymEnv scene: the kept distractors and the ball are free bodies of the full model and are drawn where they are —
    at reset (at rest where they were drawn) and after 40 steps (the balls kicked at step 10 have moved); lumped model: scenery."""
    n = 32
    cfg = _lib.default_config(_lib.ENV_KUKA_RAND)
    cfg.num_envs, cfg.seed0 = n, 6
    cfg.obs_mode, cfg.img_h, cfg.img_w = _lib.OBS_RAW_PIXELS, 64, 64
    h = _lib.Handle(cfg)
    obs = h.reset()

    def state8():
        return np.concatenate([kuka_state(h), h.get_state(_lib.F_KUKA_OBJECTS).T, h.get_state(_lib.F_KUKA_BODIES).T], axis=1)
    st0, g0 = state8(), gq(h)
    assert_images_equal(obs, raster_clib.render(8, st0, 64, 64, gripper_q=g0))
    assert_images_equal(obs, raster_clib.render(7, st0[:, :40], 64, 64, gripper_q=g0))       # at rest = the scenery rendering
    plain = raster_clib.render(4, st0[:, :10], 64, 64, gripper_q=g0)
    assert (obs != plain).reshape(n, -1).any(1).all()           # every env shows at least the ball
    out = h.rollout(40, actions=np.random.RandomState(3).randint(4, size=(40, n)).astype(np.int32))
    st = state8()
    assert np.hypot(st[:, 40 + 60] - 0.25, st[:, 40 + 61] + 0.2).min() > 1e-3           # every ball has left its drop position
    assert_images_equal(out["obs"][-1], raster_clib.render(8, st, 64, 64, gripper_q=gq(h)))
    assert (out["obs"][-1] != raster_clib.render(7, st[:, :40], 64, 64, gripper_q=gq(h))).any()
    h.close()
    cfg.kuka_model = _lib.KUKA_MODEL_LUMPED
    h = _lib.Handle(cfg)
    obs = h.reset()
    assert_images_equal(obs, raster_clib.render(7, np.concatenate([kuka_state(h), h.get_state(_lib.F_KUKA_OBJECTS).T], axis=1), 64, 64))
    with pytest.raises(_lib.SrlHipError):
        h.get_state(_lib.F_KUKA_BODIES)
    h.close()
