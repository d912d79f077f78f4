"""ctypes binding of libsrlhip.so (include/srlhip.h).

The HIP library is the product: there is NO CPU fallback.  If the shared
object is missing or no MI355X is visible, the errors raised here say so.
"""
import ctypes
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SRLHIP_LIB") or os.path.join(HERE, "libsrlhip.so")      # SRLHIP_LIB: experiment builds (profiles/probes)

# ---- constants mirrored from include/srlhip.h --------------------------------
ENV_MOBILE, ENV_MOBILE_1D, ENV_MOBILE_2TARGET, ENV_MOBILE_LINE, ENV_KUKA_BUTTON, ENV_KUKA_MOVING, ENV_KUKA_2BUTTON, ENV_KUKA_RAND = range(8)
OBS_GROUND_TRUTH, OBS_JOINTS, OBS_JOINTS_POSITION, OBS_RAW_PIXELS = range(4)
RNG_HOST, RNG_PHILOX, RNG_MT19937 = range(3)
F_POS_X, F_POS_Y, F_TARGET_X, F_TARGET_Y, F_STEP_COUNT, F_CUR_TARGET, F_LAST_REWARD, F_EP_RETURN, F_EP_LENGTH = range(9)
F_TARGET2_X, F_TARGET2_Y = 9, 10
F_LAST_RETURN, F_LAST_LENGTH, F_N_FINISHED = 11, 12, 13
F_KUKA_Q, F_KUKA_QD, F_KUKA_EE_TARGET, F_KUKA_BUTTON_Q, F_KUKA_BUTTON_POS, F_KUKA_GRIPPER, F_KUKA_COUNTERS = range(16, 23)
F_KUKA_BUTTON_XY, F_KUKA_BUTTON2_Q, F_KUKA_BUTTON2_XY, F_KUKA_GOAL, F_KUKA_OBJECTS, F_KUKA_GRIPPER_Q, F_KUKA_GRIPPER_QD, F_KUKA_BODIES, F_KUKA_IK_CROSSED = range(23, 32)
KUKA_MODEL_LUMPED, KUKA_MODEL_FULL = 0, 1          # srlhip_config.kuka_model

_FIELD_SHAPES = {
    F_POS_X: (np.float64, 1), F_POS_Y: (np.float64, 1), F_TARGET_X: (np.float64, 1), F_TARGET_Y: (np.float64, 1),
    F_STEP_COUNT: (np.int32, 1), F_CUR_TARGET: (np.int32, 1), F_LAST_REWARD: (np.float64, 1),
    F_EP_RETURN: (np.float64, 1), F_EP_LENGTH: (np.int32, 1), F_TARGET2_X: (np.float64, 1), F_TARGET2_Y: (np.float64, 1),
    F_LAST_RETURN: (np.float64, 1), F_LAST_LENGTH: (np.int32, 1), F_N_FINISHED: (np.int32, 1),
    F_KUKA_Q: (np.float64, 7), F_KUKA_QD: (np.float64, 7), F_KUKA_EE_TARGET: (np.float64, 3),
    F_KUKA_BUTTON_Q: (np.float64, 2), F_KUKA_BUTTON_POS: (np.float64, 3), F_KUKA_GRIPPER: (np.float64, 3),
    F_KUKA_COUNTERS: (np.int32, 3), F_KUKA_BUTTON_XY: (np.float64, 2), F_KUKA_BUTTON2_Q: (np.float64, 2),
    F_KUKA_BUTTON2_XY: (np.float64, 2), F_KUKA_GOAL: (np.int32, 2), F_KUKA_OBJECTS: (np.float64, 30),
    F_KUKA_GRIPPER_Q: (np.float64, 5), F_KUKA_GRIPPER_QD: (np.float64, 5), F_KUKA_BODIES: (np.float64, 66), F_KUKA_IK_CROSSED: (np.int32, 1),
}

EXPORTS = [
    "srlhip_abi_version", "srlhip_default_config", "srlhip_create", "srlhip_destroy", "srlhip_obs_dim",
    "srlhip_obs_bytes", "srlhip_action_dim", "srlhip_num_actions", "srlhip_seed", "srlhip_reset",
    "srlhip_reset_rand_count", "srlhip_step", "srlhip_step_async", "srlhip_step_wait", "srlhip_step_pending", "srlhip_set_persistent", "srlhip_rollout", "srlhip_get_state", "srlhip_set_state",
    "srlhip_device_ptr", "srlhip_render", "srlhip_episode_stats", "srlhip_episode_records", "srlhip_episode_stats_device", "srlhip_sync", "srlhip_copy_async", "srlhip_stream", "srlhip_timing_begin",
    "srlhip_timing_end", "srlhip_last_error", "srlhip_selftest_group_primitives", "srlhip_kuka_kernel", "srlhip_kuka_default_model", "srlhip_set_kuka_model", "srlhip_kuka_tree_default_model", "srlhip_set_kuka_tree_model",
    "srlhip_graph_begin", "srlhip_graph_end", "srlhip_graph_launch", "srlhip_graph_destroy",
    "srlhip_encoder_supported", "srlhip_encoder_feature_count", "srlhip_encoder_create", "srlhip_encoder_forward", "srlhip_encoder_overflow",
    "srlhip_encoder_phase_cycles",
    "srlhip_encoder_destroy", "srlhip_encoder_last_error", "srlhip_encoder_pack_bytes", "srlhip_encoder_pack", "srlhip_encoder_pack_i8_bytes", "srlhip_encoder_pack_i8", "srlhip_encoder_pack_first_layer",
]


ABI_VERSION = 5
KUKA_MODEL_DOUBLES = 138
KUKA_TREE_MODEL_DOUBLES = 510
KUKA_DETAIL_ALT_SWEEP, KUKA_DETAIL_BODY_ORDER, KUKA_DETAIL_FRICTION2 = 1, 2, 4


def kuka_tree_default_model():
    """The baked srlhip_kuka_tree_model (full 12-DoF gripper tree) as a flat float64[510] (no GPU needed)."""
    t = np.zeros(KUKA_TREE_MODEL_DOUBLES)
    rc = load().srlhip_kuka_tree_default_model(t.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0
    return t


def kuka_default_model():
    """The baked srlhip_kuka_model as a flat float64[138] (no GPU needed)."""
    t = np.zeros(KUKA_MODEL_DOUBLES)
    rc = load().srlhip_kuka_default_model(t.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0
    return t


class Config(ctypes.Structure):
    """struct srlhip_config"""
    _fields_ = [
        ("struct_size", ctypes.c_int32), ("env_kind", ctypes.c_int32), ("num_envs", ctypes.c_int32),
        ("device_id", ctypes.c_int32), ("first_env_id", ctypes.c_int32), ("is_discrete", ctypes.c_int32),
        ("random_target", ctypes.c_int32), ("force_down", ctypes.c_int32), ("shape_reward", ctypes.c_int32),
        ("action_repeat", ctypes.c_int32), ("action_joints", ctypes.c_int32), ("obs_mode", ctypes.c_int32),
        ("img_h", ctypes.c_int32), ("img_w", ctypes.c_int32), ("multi_view", ctypes.c_int32),
        ("rng_mode", ctypes.c_int32), ("auto_reset", ctypes.c_int32), ("io_device", ctypes.c_int32),
        ("kuka_model", ctypes.c_int32), ("info_bits", ctypes.c_int32),
        ("seed0", ctypes.c_int64), ("max_distance", ctypes.c_double),
    ]


class SrlHipError(RuntimeError):
    pass


_lib = None


def load():
    """dlopen libsrlhip.so; raises (never falls back) when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SrlHipError(
            "libsrlhip.so is not built ({}). Run `python -c 'import __graft_entry__ as g; g.build()'` or "
            "`make -C robotics-rl-srl_amd/csrc`. There is no CPU fallback for the env stepper.".format(LIB_PATH))
    # PyTorch ships its own libamdhip64.so; the first HIP runtime loaded serves the whole process and torch cannot see
    # the GPU through /opt/rocm's copy.  When torch is installed, let it load its runtime first (libsrlhip.so works with either).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = ctypes.CDLL(LIB_PATH)
    vp, i32 = ctypes.c_void_p, ctypes.c_int32
    if lib.srlhip_abi_version() != ABI_VERSION:       # a stale build next to a newer host package (table sizes, field ids) must not be driven
        raise SrlHipError("libsrlhip.so reports ABI version {}, this host package needs {}: rebuild it "
                          "(make -C robotics-rl-srl_amd/csrc)".format(lib.srlhip_abi_version(), ABI_VERSION))
    lib.srlhip_last_error.restype = ctypes.c_char_p
    lib.srlhip_last_error.argtypes = [vp]
    lib.srlhip_default_config.argtypes = [i32, ctypes.POINTER(Config)]
    lib.srlhip_create.argtypes = [ctypes.POINTER(Config), ctypes.POINTER(vp)]
    for name in ("srlhip_destroy", "srlhip_obs_dim", "srlhip_obs_bytes", "srlhip_action_dim", "srlhip_num_actions",
                 "srlhip_reset_rand_count", "srlhip_sync", "srlhip_timing_begin", "srlhip_step_pending"):
        getattr(lib, name).argtypes = [vp]
    lib.srlhip_seed.argtypes = [vp, vp, vp]
    lib.srlhip_reset.argtypes = [vp, vp, vp, vp]
    lib.srlhip_step.argtypes = [vp, vp, vp, vp, vp, vp]
    lib.srlhip_step_async.argtypes = [vp, vp, vp]
    lib.srlhip_step_wait.argtypes = [vp, vp, vp, vp]
    lib.srlhip_set_persistent.argtypes = [vp, i32, i32]
    lib.srlhip_rollout.argtypes = [vp, i32, vp, vp, vp, vp, vp]
    lib.srlhip_get_state.argtypes = [vp, i32, vp]
    lib.srlhip_set_state.argtypes = [vp, i32, vp]
    lib.srlhip_device_ptr.argtypes = [vp, i32, ctypes.POINTER(vp)]
    lib.srlhip_episode_stats.argtypes = [vp, vp, vp, vp]
    lib.srlhip_episode_stats_device.argtypes = [vp, vp, vp, vp]
    lib.srlhip_episode_records.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(vp)]
    lib.srlhip_selftest_group_primitives.argtypes = [i32, vp, vp, i32]
    lib.srlhip_kuka_kernel.argtypes = [vp]
    lib.srlhip_kuka_default_model.argtypes = [vp]
    lib.srlhip_set_kuka_model.argtypes = [vp, vp]
    lib.srlhip_kuka_tree_default_model.argtypes = [vp]
    lib.srlhip_set_kuka_tree_model.argtypes = [vp, vp]
    lib.srlhip_render.argtypes = [vp, vp]
    lib.srlhip_stream.argtypes = [vp, ctypes.POINTER(vp)]
    lib.srlhip_copy_async.argtypes = [vp, vp, vp, ctypes.c_size_t]
    lib.srlhip_timing_end.argtypes = [vp, ctypes.POINTER(ctypes.c_float)]
    lib.srlhip_graph_begin.argtypes = [vp]
    lib.srlhip_graph_end.argtypes = [vp, ctypes.POINTER(vp)]
    lib.srlhip_graph_launch.argtypes = [vp, vp]
    lib.srlhip_graph_destroy.argtypes = [vp]
    lib.srlhip_encoder_supported.argtypes = [i32, i32, i32]
    lib.srlhip_encoder_feature_count.argtypes = [i32, i32, i32]
    lib.srlhip_encoder_feature_count.restype = i32
    lib.srlhip_encoder_create.argtypes = [i32, i32, i32, i32, i32] + [vp] * 8 + [ctypes.POINTER(vp)]
    lib.srlhip_encoder_forward.argtypes = [vp, vp, i32, vp, vp]
    lib.srlhip_encoder_overflow.argtypes = [vp, ctypes.POINTER(i32)]
    lib.srlhip_encoder_phase_cycles.argtypes = [vp, vp, i32, vp, vp]
    lib.srlhip_encoder_destroy.argtypes = [vp]
    lib.srlhip_encoder_last_error.restype = ctypes.c_char_p
    lib.srlhip_encoder_last_error.argtypes = [vp]
    lib.srlhip_encoder_pack_bytes.restype = ctypes.c_size_t
    lib.srlhip_encoder_pack_bytes.argtypes = []
    lib.srlhip_encoder_pack.argtypes = [vp, vp, vp, vp, vp, ctypes.c_size_t, vp]
    lib.srlhip_encoder_pack_i8_bytes.restype = ctypes.c_size_t
    lib.srlhip_encoder_pack_i8_bytes.argtypes = []
    lib.srlhip_encoder_pack_i8.argtypes = [vp, vp, i32, vp, ctypes.c_size_t, vp]
    lib.srlhip_encoder_pack_first_layer.argtypes = [i32, vp, vp, vp, ctypes.c_size_t, ctypes.POINTER(ctypes.c_float)]
    _lib = lib
    return lib


def default_config(env_kind):
    cfg = Config()
    rc = load().srlhip_default_config(env_kind, ctypes.byref(cfg))
    if rc:
        raise SrlHipError("srlhip_default_config({}) -> {}".format(env_kind, rc))
    return cfg


def _ptr(a):
    if a is None:
        return None
    if isinstance(a, int):          # raw device pointer (io_device = 1)
        return ctypes.c_void_p(a)
    return a.ctypes.data_as(ctypes.c_void_p)


class Handle(object):
    """Thin object wrapper over srlhip_handle; numpy arrays in host-io mode,
    integer device pointers (e.g. torch.Tensor.data_ptr()) in device-io mode."""

    def __init__(self, cfg):
        self._lib = load()
        self._h = ctypes.c_void_p()
        self.cfg = cfg
        rc = self._lib.srlhip_create(ctypes.byref(cfg), ctypes.byref(self._h))
        if rc:
            self._h = None
            msg = self._lib.srlhip_last_error(None).decode()
            raise SrlHipError("srlhip_create failed ({}): {}".format(rc, msg))
        self.num_envs = cfg.num_envs
        self.obs_dim = self._lib.srlhip_obs_dim(self._h)
        self.obs_bytes = self._lib.srlhip_obs_bytes(self._h)
        self.action_dim = self._lib.srlhip_action_dim(self._h)
        self.num_actions = self._lib.srlhip_num_actions(self._h)
        self.reset_rand_count = self._lib.srlhip_reset_rand_count(self._h)
        # SRLHIP_KUKA_SDF=<.../pybullet_data/kuka_iiwa/kuka_with_gripper2.sdf>: take the arm model from the file the
        # reference loads (kuka.py:60) instead of the baked (recalled) table
        sdf = os.environ.get("SRLHIP_KUKA_SDF")
        if sdf and cfg.env_kind in (ENV_KUKA_BUTTON, ENV_KUKA_MOVING, ENV_KUKA_RAND):
            from . import kuka_model
            self.set_kuka_model(kuka_model.to_table(kuka_model.from_sdf(sdf)))

    def _check(self, rc, what):
        if rc:
            raise SrlHipError("{} failed ({}): {}".format(what, rc, self._lib.srlhip_last_error(self._h).decode()))

    def last_error(self):
        return self._lib.srlhip_last_error(self._h).decode()

    def close(self):
        if self._h:
            self._lib.srlhip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- allocation helpers (host io) -----------------------------------------
    def new_obs(self, T=None):
        n = self.num_envs
        lead = (n,) if T is None else (T, n)
        if self.cfg.obs_mode == OBS_RAW_PIXELS:
            ch = 6 if self.cfg.multi_view else 3
            return np.zeros(lead + (self.cfg.img_h, self.cfg.img_w, ch), np.uint8)
        return np.zeros(lead + (self.obs_dim,), np.float32)

    def action_array(self, actions, T=None):
        n = self.num_envs
        lead = (n,) if T is None else (T, n)
        if self.cfg.is_discrete:
            a = np.ascontiguousarray(actions, dtype=np.int32)
            assert a.shape == lead, (a.shape, lead)
        else:
            a = np.ascontiguousarray(actions, dtype=np.float32)
            assert a.shape == lead + (self.action_dim,), (a.shape, lead)
        return a

    # ---- ABI calls -------------------------------------------------------------
    def seed(self, seeds, mask=None):
        seeds = np.ascontiguousarray(seeds, dtype=np.int64)
        assert seeds.shape == (self.num_envs,)
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        self._check(self._lib.srlhip_seed(self._h, _ptr(m), _ptr(seeds)), "srlhip_seed")

    def reset(self, mask=None, host_rand=None, obs_out=None):
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        if host_rand is not None and not isinstance(host_rand, int):
            host_rand = np.ascontiguousarray(host_rand, dtype=np.float64)
            assert host_rand.shape == (self.num_envs, self.reset_rand_count)
        if obs_out is None and not self.cfg.io_device:
            obs_out = self.new_obs()
        self._check(self._lib.srlhip_reset(self._h, _ptr(m), _ptr(host_rand), _ptr(obs_out)), "srlhip_reset")
        return obs_out

    def step(self, actions, host_noise=None, out=None):
        if self.cfg.io_device:
            obs, rew, done = out
            self._check(self._lib.srlhip_step(self._h, _ptr(actions), _ptr(host_noise), _ptr(obs), _ptr(rew),
                                              _ptr(done)), "srlhip_step")
            return out
        a = self.action_array(actions)
        if host_noise is not None:
            host_noise = np.ascontiguousarray(host_noise, dtype=np.float64)
        if out is None:
            out = (self.new_obs(), np.zeros(self.num_envs, np.float32), np.zeros(self.num_envs, np.uint8))
        obs, rew, done = out
        self._check(self._lib.srlhip_step(self._h, _ptr(a), _ptr(host_noise), _ptr(obs), _ptr(rew), _ptr(done)),
                    "srlhip_step")
        return out

    def rollout(self, T, actions=None, want=("obs", "reward", "done", "actions"), out=None):
        n = self.num_envs
        if self.cfg.io_device:
            obs, rew, done, act = out
            self._check(self._lib.srlhip_rollout(self._h, T, _ptr(actions), _ptr(obs), _ptr(rew), _ptr(done),
                                                 _ptr(act)), "srlhip_rollout")
            return out
        a = None if actions is None else self.action_array(actions, T)
        obs = self.new_obs(T) if "obs" in want else None
        rew = np.zeros((T, n), np.float32) if "reward" in want else None
        done = np.zeros((T, n), np.uint8) if "done" in want else None
        act = None
        if a is None and "actions" in want:
            act = np.zeros((T, n), np.int32) if self.cfg.is_discrete else np.zeros((T, n, self.action_dim), np.float32)
        self._check(self._lib.srlhip_rollout(self._h, T, _ptr(a), _ptr(obs), _ptr(rew), _ptr(done), _ptr(act)),
                    "srlhip_rollout")
        return {"obs": obs, "reward": rew, "done": done, "actions": a if a is not None else act}

    def get_state(self, field):
        dtype, k = _FIELD_SHAPES[field]
        out = np.zeros((k, self.num_envs) if k > 1 else (self.num_envs,), dtype)
        self._check(self._lib.srlhip_get_state(self._h, field, _ptr(out)), "srlhip_get_state")
        return out

    def set_state(self, field, value):
        dtype, k = _FIELD_SHAPES[field]
        v = np.ascontiguousarray(value, dtype=dtype)
        assert v.shape == ((k, self.num_envs) if k > 1 else (self.num_envs,))
        self._check(self._lib.srlhip_set_state(self._h, field, _ptr(v)), "srlhip_set_state")

    def device_ptr(self, field):
        p = ctypes.c_void_p()
        self._check(self._lib.srlhip_device_ptr(self._h, field, ctypes.byref(p)), "srlhip_device_ptr")
        return p.value

    def render(self, out=None):
        """uint8 [N][H][W][C] image of the current state of every env (tile rasteriser)."""
        if self.cfg.io_device:
            self._check(self._lib.srlhip_render(self._h, _ptr(out)), "srlhip_render")
            return out
        ch = 6 if self.cfg.multi_view else 3
        if out is None:
            out = np.zeros((self.num_envs, self.cfg.img_h, self.cfg.img_w, ch), np.uint8)
        self._check(self._lib.srlhip_render(self._h, _ptr(out)), "srlhip_render")
        return out

    def set_kuka_model(self, table):
        """Install a runtime model table (srlhip_kuka_model: 138 float64, see srlhip.kuka_model) — reset() afterwards."""
        t = np.ascontiguousarray(table, dtype=np.float64)
        assert t.shape == (KUKA_MODEL_DOUBLES,)
        self._check(self._lib.srlhip_set_kuka_model(self._h, _ptr(t)), "srlhip_set_kuka_model")

    def set_kuka_tree_model(self, table):
        """Install a full-model table (srlhip_kuka_tree_model: 510 float64) on a KUKA_MODEL_FULL handle — reset() afterwards."""
        t = np.ascontiguousarray(table, dtype=np.float64)
        assert t.shape == (KUKA_TREE_MODEL_DOUBLES,)
        self._check(self._lib.srlhip_set_kuka_tree_model(self._h, _ptr(t)), "srlhip_set_kuka_tree_model")

    def kuka_kernel(self):
        """'tree' (full model, 16 lanes per env), 'group' (lumped model, 16 lanes per env) or 'lane' (lumped, one lane per env):
        the kernel that steps this Kuka batch."""
        rc = self._lib.srlhip_kuka_kernel(self._h)
        if rc < 0:
            raise SrlHipError("srlhip_kuka_kernel: not a Kuka handle")
        return {0: "lane", 1: "group", 2: "tree"}[rc]

    def kuka_model_name(self):
        return "full 12-DoF gripper tree" if self.cfg.kuka_model == KUKA_MODEL_FULL else "lumped gripper (7 DoF)"

    def episode_stats_device(self, last_return=0, last_length=0, n_finished=0):
        """Enqueue-only: float32 returns / int32 lengths / counts of the last finished episodes into DEVICE buffers
        (raw pointers, 0 = skip) on the handle's stream."""
        self._check(self._lib.srlhip_episode_stats_device(self._h, last_return or None, last_length or None, n_finished or None),
                    "srlhip_episode_stats_device")

    def episode_records(self):
        """Zero-copy numpy views (last_return float64 [n], last_length int32 [n]) of the mapped host block a host-pointer handle's
        kernels write Monitor's (r, l) into: read entry i after a step / rollout call in which env i reported done."""
        r, l = ctypes.c_void_p(), ctypes.c_void_p()
        self._check(self._lib.srlhip_episode_records(self._h, ctypes.byref(r), ctypes.byref(l)), "srlhip_episode_records")
        n = self.num_envs
        ret = np.ctypeslib.as_array(ctypes.cast(r, ctypes.POINTER(ctypes.c_double)), shape=(n,))
        length = np.ctypeslib.as_array(ctypes.cast(l, ctypes.POINTER(ctypes.c_int32)), shape=(n,))
        ret.flags.writeable = length.flags.writeable = False
        return ret, length

    def step_fn(self, actions, host_noise=None):
        """A closure for per-step loops on a host-pointer handle: the output arrays, their ctypes pointers and the C entry point are
        bound once, so a step costs one foreign call.  `actions` is the caller's int32 / float32 array that it refills in place
        before every call; returns (call, obs, reward, done) — call() steps and returns the library's status code."""
        assert not self.cfg.io_device
        a = self.action_array(actions)
        assert a is actions, "step_fn needs the final contiguous int32 / float32 action array"
        obs, rew, done = self.new_obs(), np.zeros(self.num_envs, np.float32), np.zeros(self.num_envs, np.uint8)
        fn, hh = self._lib.srlhip_step, self._h
        pa, pn, po, pr, pd = _ptr(a), _ptr(host_noise), _ptr(obs), _ptr(rew), _ptr(done)
        return (lambda: fn(hh, pa, pn, po, pr, pd)), obs, rew, done

    def step_split_fn(self, actions, obs, rew, done):
        """The two halves of a per-step loop on a host-pointer handle, bound once to the CALLER's arrays (contiguous, e.g. this
        shard's slices of global planes): go() = srlhip_step_async on `actions` (refilled in place before every call), collect() =
        srlhip_step_wait into obs / rew / done.  Both return the library's status code."""
        assert not self.cfg.io_device
        assert self.action_array(actions) is actions, "step_split_fn needs the final contiguous int32 / float32 action array"
        assert obs.flags.c_contiguous and obs.nbytes == self.obs_bytes * self.num_envs, (obs.shape, obs.dtype)
        assert rew.flags.c_contiguous and rew.dtype == np.float32 and rew.shape == (self.num_envs,)
        assert done.flags.c_contiguous and done.dtype == np.uint8 and done.shape == (self.num_envs,)
        go, wait, hh = self._lib.srlhip_step_async, self._lib.srlhip_step_wait, self._h
        pa, po, pr, pd = _ptr(actions), _ptr(obs), _ptr(rew), _ptr(done)
        keep = (actions, obs, rew, done)                     # the pointers above borrow these arrays
        return (lambda: go(hh, pa, None)), (lambda _keep=keep: wait(hh, po, pr, pd))

    def step_async(self, actions, host_noise=None):
        """srlhip_step_async: enqueue one step of a host-pointer handle (returns before the GPU has run it)"""
        a = self.action_array(actions)
        if host_noise is not None:
            host_noise = np.ascontiguousarray(host_noise, dtype=np.float64)
        self._check(self._lib.srlhip_step_async(self._h, _ptr(a), _ptr(host_noise)), "srlhip_step_async")

    def step_wait(self, out=None):
        """srlhip_step_wait: (obs, reward, done) of the step srlhip_step_async enqueued"""
        if out is None:
            out = (self.new_obs(), np.zeros(self.num_envs, np.float32), np.zeros(self.num_envs, np.uint8))
        obs, rew, done = out
        self._check(self._lib.srlhip_step_wait(self._h, _ptr(obs), _ptr(rew), _ptr(done)), "srlhip_step_wait")
        return out

    def step_pending(self):
        return self._lib.srlhip_step_pending(self._h) == 1

    def set_persistent(self, on=True, park_us=0):
        """srlhip_set_persistent: per-step calls without a launch per step (a resident kernel takes its steps through mapped memory).
        Raises SrlHipError (ENOTSUP) for handles that have no persistent form."""
        self._check(self._lib.srlhip_set_persistent(self._h, int(bool(on)), int(park_us)), "srlhip_set_persistent")

    def episode_stats(self):
        n = self.num_envs
        ret, length, fin = np.zeros(n, np.float64), np.zeros(n, np.int32), np.zeros(n, np.int32)
        self._check(self._lib.srlhip_episode_stats(self._h, _ptr(ret), _ptr(length), _ptr(fin)),
                    "srlhip_episode_stats")
        return ret, length, fin

    def sync(self):
        self._check(self._lib.srlhip_sync(self._h), "srlhip_sync")

    def copy_async(self, dst, src, nbytes):
        """srlhip_copy_async: raw pointers (device, or pinned host), enqueued on the handle's stream"""
        self._check(self._lib.srlhip_copy_async(self._h, ctypes.c_void_p(dst), ctypes.c_void_p(src), ctypes.c_size_t(nbytes)), "srlhip_copy_async")

    def stream(self):
        p = ctypes.c_void_p()
        self._check(self._lib.srlhip_stream(self._h, ctypes.byref(p)), "srlhip_stream")
        return p.value

    # ---- HIP graphs (device-io handles): capture a step sequence once, replay it with one launch per step
    def graph_begin(self):
        self._check(self._lib.srlhip_graph_begin(self._h), "srlhip_graph_begin")

    def graph_end(self):
        g = ctypes.c_void_p()
        self._check(self._lib.srlhip_graph_end(self._h, ctypes.byref(g)), "srlhip_graph_end")
        return g

    def graph_launch(self, g):
        self._check(self._lib.srlhip_graph_launch(self._h, g), "srlhip_graph_launch")

    def graph_destroy(self, g):
        self._lib.srlhip_graph_destroy(g)

    def timing_begin(self):
        self._check(self._lib.srlhip_timing_begin(self._h), "srlhip_timing_begin")

    def timing_end(self):
        ms = ctypes.c_float()
        self._check(self._lib.srlhip_timing_end(self._h, ctypes.byref(ms)), "srlhip_timing_end")
        return ms.value


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def encoder_supported(img_h, img_w, n_channels):
    return bool(load().srlhip_encoder_supported(int(img_h), int(img_w), int(n_channels)))


def encoder_pack_first_layer(conv1_w, conv1_b):
    """Host-only: (layer 1's packed MFMA B-operand image of the layered encoder as float16 words, its power-of-two pre-scale)
    for 3- or 6-channel frames — csrc/encoder_general.hip's packer, as srlhip_encoder_create uploads it."""
    lib = load()
    w1, b1 = _f32(conv1_w), _f32(conv1_b)
    c = w1.shape[1]
    assert w1.shape == (64, c, 7, 7) and c in (3, 6) and b1.shape == (64,)
    out = np.zeros(2 * (14 if c == 3 else 28) * 1024, np.float16)
    scale = ctypes.c_float()
    rc = lib.srlhip_encoder_pack_first_layer(c, _ptr(w1), _ptr(b1), _ptr(out), ctypes.c_size_t(out.nbytes), ctypes.byref(scale))
    if rc:
        raise SrlHipError("srlhip_encoder_pack_first_layer failed ({})".format(rc))
    return out, float(scale.value)


def encoder_feature_count(img_h, img_w, n_channels):
    """inputs of the encoder's fully connected layer for this frame shape, 0 when the shape is not covered"""
    return int(load().srlhip_encoder_feature_count(int(img_h), int(img_w), int(n_channels)))


def encoder_pack(conv1_w, conv1_b, conv2_w, conv3_w):
    """Host-only: (packed MFMA B-operand image as float16 words, the three per-layer power-of-two weight scales)
    exactly as srlhip_encoder_create uploads them."""
    lib = load()
    out = np.zeros(lib.srlhip_encoder_pack_bytes() // 2, np.float16)
    w1, b1, w2, w3 = _f32(conv1_w), _f32(conv1_b), _f32(conv2_w), _f32(conv3_w)
    assert w1.shape == (64, 3, 7, 7) and b1.shape == (64,) and w2.shape == (64, 64, 3, 3) and w3.shape == (64, 64, 3, 3)
    scales = np.zeros(3, np.float32)
    rc = lib.srlhip_encoder_pack(_ptr(w1), _ptr(b1), _ptr(w2), _ptr(w3), _ptr(out), out.nbytes, _ptr(scales))
    if rc:
        raise SrlHipError("srlhip_encoder_pack failed ({})".format(rc))
    return out, scales


def encoder_pack_i8(conv1_w, conv1_b, zero_slot=0):
    """Host-only: (int8 digits [2][7][3][64][16], float32 [64] per-channel 256 / scale) of the int8 layer 1 — zero_slot 0: the fused
    64x64x3 kernel's layout, 7: the layered kernels' (3-channel frames of any other size)."""
    lib = load()
    w1, b1 = _f32(conv1_w), _f32(conv1_b)
    assert w1.shape == (64, 3, 7, 7) and b1.shape == (64,)
    out = np.zeros(lib.srlhip_encoder_pack_i8_bytes(), np.int8)
    inv = np.zeros(64, np.float32)
    rc = lib.srlhip_encoder_pack_i8(_ptr(w1), _ptr(b1), int(zero_slot), _ptr(out), out.nbytes, _ptr(inv))
    if rc:
        raise SrlHipError("srlhip_encoder_pack_i8 failed ({})".format(rc))
    return out.reshape(2, 7, 3, 64, 16), inv


class Encoder(object):
    """srlhip_encoder_handle: the CustomCNN forward on device-resident uint8 frames — one fused kernel for 64x64x3
    (csrc/encoder.hip), the layered kernels of csrc/encoder_general.hip for any other shape.
    Weights: float32 arrays in torch layout with the BatchNorms already folded (see include/srlhip.h); fc_w is
    [state_dim][64 * pooled cells] in torch's flatten order."""

    def __init__(self, device_id, img_shape, n_channels, state_dim, conv1, conv2, conv3, fc):
        self._lib = load()
        self._e = ctypes.c_void_p()
        arrs = [_f32(a) for pair in (conv1, conv2, conv3, fc) for a in pair]
        assert arrs[0].shape == (64, n_channels, 7, 7) and arrs[2].shape == (64, 64, 3, 3) and arrs[4].shape == (64, 64, 3, 3)
        assert arrs[6].shape == (state_dim, encoder_feature_count(img_shape[0], img_shape[1], n_channels)) and arrs[7].shape == (state_dim,)
        rc = self._lib.srlhip_encoder_create(int(device_id), int(img_shape[0]), int(img_shape[1]), int(n_channels),
                                             int(state_dim), *[_ptr(a) for a in arrs], ctypes.byref(self._e))
        if rc:
            self._e = None
            raise SrlHipError("srlhip_encoder_create failed ({}): {}".format(
                rc, self._lib.srlhip_encoder_last_error(None).decode()))
        self.state_dim = int(state_dim)

    def _check(self, rc, what):
        if rc:
            raise SrlHipError("{} failed ({}): {}".format(what, rc, self._lib.srlhip_encoder_last_error(self._e).decode()))

    def forward(self, images_ptr, n, states_ptr, stream=None):
        """images_ptr / states_ptr: raw device pointers (e.g. torch.Tensor.data_ptr()); enqueue-only on `stream`."""
        self._check(self._lib.srlhip_encoder_forward(self._e, _ptr(images_ptr), int(n), _ptr(states_ptr),
                                                     ctypes.c_void_p(stream) if stream else None), "srlhip_encoder_forward")

    PHASES = ("unpack", "layer1", "barrier1", "layer2_kloop", "layer2_epilogue", "barrier2", "layer3", "fc", "turnaround")

    def phase_cycles(self, images_ptr, n, states_ptr):
        """Diagnostic forward: {phase: shader cycles} of workgroup 0 (slowest wave, averaged over its first frames)."""
        c = np.zeros(9, np.int64)
        self._check(self._lib.srlhip_encoder_phase_cycles(self._e, _ptr(images_ptr), int(n), _ptr(states_ptr), _ptr(c)),
                    "srlhip_encoder_phase_cycles")
        return dict(zip(self.PHASES, c.tolist()))

    def overflow(self):
        flag = ctypes.c_int32()
        self._check(self._lib.srlhip_encoder_overflow(self._e, ctypes.byref(flag)), "srlhip_encoder_overflow")
        return bool(flag.value)

    def close(self):
        if self._e:
            self._lib.srlhip_encoder_destroy(self._e)
            self._e = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
