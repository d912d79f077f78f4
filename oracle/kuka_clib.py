"""ctypes front end of the plain-C Kuka oracle (oracle/kuka_oracle.c).
TEST INFRASTRUCTURE ONLY — see oracle/__init__.py."""
import ctypes
import os
import time

import numpy as np

from . import clib

RNG_PHILOX, RNG_MT19937 = 1, 2


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _lib():
    lib = clib.lib()
    lib.kuka_oracle_rollout.argtypes = (
        [ctypes.c_int] * 6 + [ctypes.c_double] + [ctypes.c_int] * 5 + [ctypes.c_void_p] * 14)
    lib.kuka_oracle_aba.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_double, ctypes.c_void_p]
    lib.kuka_oracle_wrapper_step.argtypes = ([ctypes.c_void_p] * 3 + [ctypes.c_int] * 4 + [ctypes.c_double] +
                                             [ctypes.c_void_p] * 2)
    return lib


def set_moving(flag):
    """Select KukaMovingButtonGymEnv semantics for the following rollout / trace / wrapper calls."""
    _lib().kuka_oracle_set_moving(int(bool(flag)))


VARIANT_BUTTON, VARIANT_MOVING, VARIANT_TWO, VARIANT_RAND = 0, 1, 2, 3


def set_variant(variant):
    """0 KukaButtonGymEnv, 1 KukaMovingButtonGymEnv, 2 Kuka2ButtonGymEnv, 3 KukaRandButtonGymEnv for the following calls."""
    _lib().kuka_oracle_set_variant(int(variant))


MODEL_DOUBLES = 138
MODEL_FIELDS = (("joint_xyz", (7, 3)), ("joint_rpy", (7, 3)), ("joint_lower", (7,)), ("joint_upper", (7,)), ("joint_damping", ()),
                ("mass", (7,)), ("com", (7, 3)), ("inertia", (7, 3)), ("ee_point", (3,)), ("gripper_point", (3,)), ("sphere", (6, 4)),
                ("table_top_z", ()), ("button_base_z", ()))


def model_to_dict(table):
    out, k = {}, 0
    for name, shape in MODEL_FIELDS:
        n = int(np.prod(shape)) if shape else 1
        out[name] = np.array(table[k:k + n]).reshape(shape) if shape else float(table[k])
        k += n
    assert k == MODEL_DOUBLES
    return out


def model_to_table(model):
    return np.concatenate([np.asarray(model[name], dtype=np.float64).reshape(-1) for name, _ in MODEL_FIELDS])


def get_model():
    """The recalled part of the Kuka model (link frames, inertial parameters, limits, collision spheres, table / button
    heights) as the oracle currently integrates it: a dict of arrays in the layout of srlhip_kuka_model."""
    t = np.zeros(MODEL_DOUBLES)
    clib.lib().kuka_oracle_get_model(_p(t))
    return model_to_dict(t)


def set_model(model):
    """Install a model table (dict as returned by get_model(), or the flat 138 doubles) in the physics AND the raster oracle."""
    t = np.ascontiguousarray(model if not isinstance(model, dict) else model_to_table(model), dtype=np.float64)
    assert t.shape == (MODEL_DOUBLES,)
    clib.lib().kuka_oracle_set_model(_p(t))
    clib.lib().raster_oracle_set_model(_p(t))


def set_full(flag):
    """False: gripper lumped rigidly into link_7 (7 DoF, the rounds 1-2 model).  True: the full kuka_with_gripper2 tree — 12 DoF
    with every gripper joint motor of kuka.py:172-187, 16 collision spheres on links 5..11, one friction row per contact."""
    clib.lib().kuka_oracle_set_full(int(bool(flag)))


def is_full():
    return bool(clib.lib().kuka_oracle_get_full())


def ndof():
    return 12 if is_full() else 7


def get_tree_model():
    """flat float64 image of the tree model table in use (layout of srlhip_kuka_tree_model)"""
    t = np.zeros(clib.lib().kuka_oracle_tree_doubles())
    clib.lib().kuka_oracle_get_tree_model(_p(t))
    return t


def set_tree_model(table):
    t = np.ascontiguousarray(table, dtype=np.float64)
    assert t.shape == (clib.lib().kuka_oracle_tree_doubles(),)
    clib.lib().kuka_oracle_set_tree_model(_p(t))


def margins_reset():
    """Start a new flag-margin record (oracle/kuka_oracle.c: margin probe); rollout() calls accumulate into it."""
    clib.lib().kuka_oracle_margins_reset()


def margins():
    """{"contact_button", "contact_table", "max_distance", "any_contact"}: the smallest |value - threshold| the quantities behind the
    discrete reward / done outputs reached in the rollouts since margins_reset()."""
    out = np.zeros(4)
    clib.lib().kuka_oracle_margins_get(_p(out))
    return dict(zip(("contact_button", "contact_table", "max_distance", "any_contact"), out))


def body_trace(n):
    """KukaRandButton: arm the free-body trace of the NEXT rollout() call; returns the [n][11][7] array it fills (x y z vx vy vz on per
    body: ten distractors in draw order, then the ball) with every env's state after the last step."""
    out = np.zeros((n, 11, 7))
    clib.lib().kuka_oracle_set_body_trace(_p(out))
    return out


def body_trace_off():
    clib.lib().kuka_oracle_set_body_trace(None)


def rb_drop_check():
    """Largest distance of any free body, after the reference's literal drop (0.1 m / 0.3 m above Z_TABLE, then the 500 settle steps),
    from the rest state reset() places it in."""
    clib.lib().kuka_oracle_rb_drop_check.restype = ctypes.c_double
    return float(clib.lib().kuka_oracle_rb_drop_check())


def _pad(x):
    x = np.asarray(x, dtype=np.float64).reshape(-1)
    out = np.zeros(12)
    out[:len(x)] = x
    return out


def aba(q, qd, tau, gz=-10.0):
    q, qd, tau = (_pad(x) for x in (q, qd, tau))
    out = np.zeros(12)
    _lib().kuka_oracle_aba(_p(q), _p(qd), _p(tau), gz, _p(out))
    return out[:ndof()]


def minv(q):
    q, n = _pad(q), ndof()
    out = np.zeros((n, n))
    _lib().kuka_oracle_minv(_p(q), _p(out))
    return out


def fk(q):
    q, n = _pad(q), ndof()
    R, p = np.zeros((n, 3, 3)), np.zeros((n, 3))
    _lib().kuka_oracle_fk(_p(q), _p(R), _p(p))
    return R, p


def ik(q, target):
    q = _pad(q)
    target = np.ascontiguousarray(target, dtype=np.float64)
    out = np.zeros(7)
    _lib().kuka_oracle_ik(_p(q), _p(target), _p(out))
    return out


def settled(random_target=False, action_joints=False):
    out = np.zeros(22)
    _lib().kuka_oracle_settled(int(random_target), int(action_joints), _p(out))
    return {"q": out[:7], "qd": out[7:14], "ee_target": out[14:17], "button": out[17:19], "gripper": out[19:22]}


def rollout(seeds, T, actions=None, is_discrete=True, action_joints=False, random_target=False, force_down=True,
            shape_reward=False, action_repeat=1, max_distance=0.8, obs_mode=0, rng_mode=RNG_MT19937, auto_reset=True,
            trace=True, aux=False, ik_trace=False):
    """aux: also return q_all [T][n][12] (every DoF of the model in use) and rows [T][n][3] (contact-normal rows, friction rows + 1000 x joint-limit rows, arm <-> free-body contact rows).
    ik_trace: also ik_det [T][n] (det(J^T J + damping I) of the step's IK solve), ik_crossed [T][n] (the episode's sticky conditioning bit after
    the step: some IK solve of the episode had det < KM_IK_CROSS_DET), ik_final [n][2] (the bit at the end, env-steps taken with it set)."""
    seeds = np.ascontiguousarray(seeds, dtype=np.int64)
    n = len(seeds)
    od = {0: 3, 1: 14, 2: 17}[obs_mode]
    adim = 1 if is_discrete else (7 if action_joints else 3)
    keys, lens = clib.mt_keys(seeds) if rng_mode == RNG_MT19937 else (np.zeros((n, 2), np.uint32), np.zeros(n, np.int32))
    out = {
        "obs0": np.zeros((n, od), np.float32), "obs": np.zeros((T, n, od), np.float32),
        "reward": np.zeros((T, n), np.float32), "reward64": np.zeros((T, n)), "done": np.zeros((T, n), np.uint8),
        "q": np.zeros((T, n, 7)) if trace else None, "gripper": np.zeros((T, n, 3)) if trace else None,
        "final_state": np.zeros((n, 40)), "ep_stats": np.zeros((n, 3)),
    }
    if ik_trace:
        out["ik_det"], out["ik_crossed"], out["ik_final"] = np.zeros((T, n)), np.zeros((T, n), np.uint8), np.zeros((n, 2), np.int32)
        clib.lib().kuka_oracle_set_ik_trace(_p(out["ik_det"]), _p(out["ik_crossed"]), _p(out["ik_final"]))
    if aux:
        out["q_all"] = np.zeros((T, n, 12))
        out["rows"] = np.zeros((T, n, 3), np.int32)
        clib.lib().kuka_oracle_set_aux_trace(_p(out["q_all"]), _p(out["rows"]))
    act_out = None
    if actions is None:
        act_out = np.zeros((T, n), np.int32) if is_discrete else np.zeros((T, n, adim), np.float32)
    else:
        actions = np.ascontiguousarray(actions, dtype=np.int32 if is_discrete else np.float32)
        assert actions.shape == ((T, n) if is_discrete else (T, n, adim)), actions.shape
    rc = _lib().kuka_oracle_rollout(
        int(is_discrete), int(action_joints), int(random_target), int(force_down), int(shape_reward),
        int(action_repeat), float(max_distance), int(obs_mode), int(rng_mode), int(auto_reset), n, int(T),
        _p(seeds), _p(keys), _p(lens), _p(actions), _p(out["obs0"]), _p(out["obs"]), _p(out["reward"]),
        _p(out["reward64"]), _p(out["done"]), _p(act_out), _p(out["q"]), _p(out["gripper"]),
        _p(out["final_state"]), _p(out["ep_stats"]))
    if ik_trace:
        clib.lib().kuka_oracle_set_ik_trace(None, None, None)
    if aux:
        clib.lib().kuka_oracle_set_aux_trace(None, None)
    assert rc == 0
    out["actions"] = actions if actions is not None else act_out
    return out


def wrapper_step(state, gripper, button_pos, contact_button, contact_table, shape_reward=False, is_discrete=True,
                 max_distance=0.8):
    st = np.ascontiguousarray(state, dtype=np.float64)
    st = np.concatenate([st, np.zeros(8 - len(st))])
    g = np.ascontiguousarray(gripper, dtype=np.float64)
    b = np.ascontiguousarray(button_pos, dtype=np.float64)
    reward, done = ctypes.c_double(), ctypes.c_int()
    _lib().kuka_oracle_wrapper_step(_p(st), _p(g), _p(b), int(contact_button), int(contact_table), int(shape_reward),
                                    int(is_discrete), float(max_distance), ctypes.byref(reward), ctypes.byref(done))
    return st[:4].copy(), reward.value, bool(done.value)


def wrapper_step_two(state, gripper, all_pos, contact_goal, contact_table, shape_reward=False, max_distance=2.0):
    """Kuka2ButtonGymEnv bookkeeping probe; state = counter n_contacts[0] n_outside terminated goal_id n_contacts[1]."""
    st = np.ascontiguousarray(state, dtype=np.float64)
    st = np.concatenate([st, np.zeros(8 - len(st))])
    g = np.ascontiguousarray(gripper, dtype=np.float64)
    b = np.ascontiguousarray(all_pos, dtype=np.float64).reshape(6)
    reward, done = ctypes.c_double(), ctypes.c_int()
    lib = _lib()
    lib.kuka_oracle_wrapper_step_two.argtypes = ([ctypes.c_void_p] * 3 + [ctypes.c_int] * 3 + [ctypes.c_double] +
                                                 [ctypes.c_void_p] * 2)
    lib.kuka_oracle_wrapper_step_two(_p(st), _p(g), _p(b), int(contact_goal), int(contact_table), int(shape_reward),
                                     float(max_distance), ctypes.byref(reward), ctypes.byref(done))
    return st[:6].copy(), reward.value, bool(done.value)


def last_buttons():
    """button base positions (b1x b1y b2x b2y) drawn by the reset() of the last command_trace call"""
    out = np.zeros(4)
    _lib().kuka_oracle_last_buttons(_p(out))
    return out


def last_objects():
    """KukaRandButton: (x, y, present) of the ten distractors drawn by the reset() of the last command_trace call"""
    out = np.zeros((10, 3))
    _lib().kuka_oracle_last_objects(_p(out))
    return out


def command_trace(seed, T, actions, is_discrete=True, action_joints=False, random_target=False, force_down=True):
    """IK targets / joint targets issued during reset() (5 init actions) and each step of one env."""
    keys, lens = clib.mt_keys([seed])
    actions = np.ascontiguousarray(actions, dtype=np.int32 if is_discrete else np.float32)
    ee, jt = np.zeros((T + 5, 3)), np.zeros((T + 5, 7))
    n_reset = ctypes.c_int()
    lib = _lib()
    lib.kuka_oracle_command_trace.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 4
    n = lib.kuka_oracle_command_trace(int(is_discrete), int(action_joints), int(random_target), int(force_down),
                                      _p(keys), int(lens[0]), int(T), _p(actions), _p(ee), _p(jt), ctypes.byref(n_reset))
    return {"reset_ee": ee[:n_reset.value], "reset_jt": jt[:n_reset.value], "ee": ee[n_reset.value:n_reset.value + n],
            "jt": jt[n_reset.value:n_reset.value + n], "n_steps": n}


class OracleEnv:
    """One KukaButtonGymEnv the way a SubprocVecEnv worker holds it (literal 505-step reset): bench cpu_baseline only."""

    def __init__(self, seed, rng_mode=RNG_MT19937, is_discrete=True, random_target=False, force_down=True,
                 shape_reward=False, action_repeat=1, max_distance=0.8, obs_mode=0):
        lib = _lib()
        lib.kuka_oracle_env_new.restype = ctypes.c_void_p
        lib.kuka_oracle_env_new.argtypes = [ctypes.c_int] * 5 + [ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_int64,
                                                                 ctypes.c_void_p, ctypes.c_int]
        lib.kuka_oracle_env_reset.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        lib.kuka_oracle_env_step.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        lib.kuka_oracle_env_step.restype = ctypes.c_double
        lib.kuka_oracle_env_free.argtypes = [ctypes.c_void_p]
        keys, lens = clib.mt_keys([seed])
        self._lib = lib
        self._obs = np.zeros({0: 3, 1: 14, 2: 17}[obs_mode], np.float32)
        self._done = ctypes.c_int()
        self._h = lib.kuka_oracle_env_new(int(is_discrete), int(random_target), int(force_down), int(shape_reward),
                                          int(action_repeat), float(max_distance), int(obs_mode), int(rng_mode), int(seed),
                                          _p(keys), int(lens[0]))

    def reset(self):
        self._lib.kuka_oracle_env_reset(self._h, _p(self._obs))
        return self._obs.copy()

    def step(self, action):
        r = self._lib.kuka_oracle_env_step(self._h, -1 if action is None else int(action), _p(self._obs), ctypes.byref(self._done))
        return self._obs.copy(), r, bool(self._done.value)

    def close(self):
        if self._h:
            self._lib.kuka_oracle_env_free(self._h)
            self._h = None


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(budget_s=10.0, n_envs=None, subproc=True, full=True):
    was_full = is_full()
    set_full(full)           # the model the GPU path integrates by default (forked subproc workers inherit it)
    try:
        res = _cpu_baseline(budget_s, n_envs, subproc)
        res["kuka_model"] = "full 12-DoF gripper tree" if full else "lumped gripper (7 DoF)"
        return res
    finally:
        set_full(was_full)


def _cpu_baseline(budget_s=10.0, n_envs=None, subproc=True):
    """bench.py's cpu_baseline leg (rank 0, N=1): the oracle timed on the host cores, two ways.
    value   = OpenMP over envs, T = 2048 steps per env with auto-reset inside (SURVEY 8(d) rollout length); the 500 RNG-free
              settle steps are integrated once per pass and cached, a reset then costs its 5 init-action steps — the most
              generous reading for the CPU (the reference re-simulates all 505 steps and renders 224x224 every step).
    subproc = the reference's own vectorisation protocol (SubprocVecEnv: one worker process per env, Pipe + pickle per
              step, auto-reset in the worker with the literal 505-step reset), num_cpu = os.cpu_count()."""
    threads = os.cpu_count() or 1
    n = n_envs or max(64, min(4096, 4 * threads))
    T = 2048
    rollout(np.arange(min(n, 64)), 2, actions=None, rng_mode=RNG_PHILOX, trace=False)      # warm-up (page in, OpenMP pool)
    t0 = time.perf_counter()
    reps, resets = 0, 0
    while True:
        out = rollout(np.arange(n) + reps * n, T, actions=None, rng_mode=RNG_PHILOX, trace=False)
        resets += int(out["ep_stats"][:, 2].sum())
        reps += 1
        if time.perf_counter() - t0 >= budget_s:
            break
    dt = time.perf_counter() - t0
    res = {"value": reps * T * n / dt, "unit": "env-steps/s", "cores": threads, "kind": "port", "cpu_model": cpu_model(),
           "physics_steps_per_s": (reps * (T * n + 5 * n + 500) + 5 * resets) / dt,
           "sample": "oracle/kuka_oracle.c (OpenMP over envs, {} threads), {} envs x {} steps x {} passes with auto-reset inside "
                     "({} episode ends; settled state cached per pass, a reset = 5 init-action steps), physics only "
                     "(no rendering)".format(threads, n, T, reps, resets)}
    if subproc:
        from . import subproc_baseline
        try:
            res["subproc"] = subproc_baseline.kuka_subproc_fps(num_cpu=threads)
        except Exception as exc:           # a failing worker pool must not sink the bench line
            res["subproc"] = {"value": None, "error": repr(exc)}
    return res
