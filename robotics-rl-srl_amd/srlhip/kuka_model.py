"""The Kuka-button model as data, and bookkeeping constants of the Kuka stepper used by bench.py's roofline line.

`srlhip_kuka_model` (include/srlhip.h) holds every constant of the arm / scene model that the reference does NOT pin in its
own source — link frames, inertial parameters, joint limits and damping of pybullet_data/kuka_iiwa/kuka_with_gripper2.sdf
(kuka.py:60), the IK / gripper reference points, the gripper's collision spheres, the table / button-base heights — as 138
float64.  The library ships a baked table (recalled from upstream, SURVEY App. B.4: PARITY UNPINNED); `from_sdf()` builds
the table from the real files when they are available, and `Handle.set_kuka_model()` installs it.

FLOPS_PER_ENV_STEP counts the float64 operations of one physics step as
implemented in csrc/kuka_core.hpp (action_repeat = 1), an FMA = 2 flops:
  150 Gauss-Seidel sweeps x (7 arm-motor rows x (6 + 14) + 2 button rows x 8)  = 23 400
  M^-1: 7 unit-torque sweeps x 7 links x ~2 x 26                               =  2 500
  ABA backward (7 x ~330) + forward (7 x ~60)                                  =  2 700
  FK x2, IK (J^T J, LDL^T solve), 7 sincos (~40 each), contacts, rows          =  2 400
"""
import numpy as np

FLOPS_PER_ENV_STEP = 3.1e4

# The lane-group kernel (csrc/kuka_group.hpp) integrates the same model with a different row set: impulse-space rows scaled
# to [0, 1] (a row update = 1 add + clamp + one FMA into each of the OTHER coupled rows), M by CRBA and M^-1 by Gauss-Jordan
# instead of seven ABA sweeps.  Algorithmic float64 operations per env-step (the work of ONE env, not of its 16 lanes; FMA = 2,
# clamp = 2), lumped-gripper model, free path (no contact row):
#   150 sweeps x (7 arm rows x (1 + 2 + 6 x 2) + 3 button rows x (1 + 2 + 2 x 2))                    = 18 900
#   Gauss-Jordan inverse of the 7x7 mass matrix: 7 pivots x 7 rows x 7 columns x 2                     =    690
#   CRBA: composite inertias (suffix sums), Ic S per link, 28 entries x 6-term dot products            =    620
#   RNEA in world coordinates: per-link force ~150 x 7, prefix / suffix sums 27 quantities x 7 x 4 x 2 =  2 560
#   FK (7 frame compositions x 63) + 7 sincos (~40 each)                                               =    720
#   IK: Jacobian, J^T J (28 x 12), 7x7 Gauss-Jordan solve, orientation error                           =    990
#   row setup, 6 sphere-cylinder candidates, env logic, Philox                                         =    530
FLOPS_PER_ENV_STEP_GROUP = 2.5e4

MODEL_DOUBLES = 138
MODEL_FIELDS = (("joint_xyz", (7, 3)), ("joint_rpy", (7, 3)), ("joint_lower", (7,)), ("joint_upper", (7,)), ("joint_damping", ()),
                ("mass", (7,)), ("com", (7, 3)), ("inertia", (7, 3)), ("ee_point", (3,)), ("gripper_point", (3,)), ("sphere", (6, 4)),
                ("table_top_z", ()), ("button_base_z", ()))


def to_dict(table):
    """flat float64[138] -> {field: array}"""
    table = np.asarray(table, dtype=np.float64).reshape(-1)
    assert table.shape == (MODEL_DOUBLES,)
    out, k = {}, 0
    for name, shape in MODEL_FIELDS:
        n = int(np.prod(shape)) if shape else 1
        out[name] = table[k:k + n].reshape(shape).copy() if shape else float(table[k])
        k += n
    return out


def to_table(model):
    """{field: array} -> flat float64[138]"""
    t = np.concatenate([np.asarray(model[name], dtype=np.float64).reshape(-1) for name, _ in MODEL_FIELDS])
    assert t.shape == (MODEL_DOUBLES,)
    return t


def default():
    """The baked table of the library (no GPU needed)."""
    from . import _lib
    return to_dict(_lib.kuka_default_model())


def _rpy_matrix(rpy):
    r, p, y = rpy
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    return np.array([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
                     [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                     [-sp, cp * sr, cp * cr]])


def _matrix_rpy(R):
    p = -np.arcsin(np.clip(R[2, 0], -1.0, 1.0))
    if abs(np.cos(p)) > 1e-9:
        return np.array([np.arctan2(R[2, 1], R[2, 2]), p, np.arctan2(R[1, 0], R[0, 0])])
    return np.array([np.arctan2(-R[1, 2], R[1, 1]), p, 0.0])


def from_sdf(sdf_path, base=None):
    """Arm part of the table from kuka_iiwa/kuka_with_gripper2.sdf (or any sdf / urdf-like model whose first 7 revolute joints
    form a serial chain turning about the child frames' z): link poses -> joint origins in the parent frame, inertial
    pose / mass / diagonal inertia per link, joint limits and damping.  Fields the file does not carry (ee_point,
    gripper_point, collision spheres, table / button heights) are taken from `base` (default: the baked table).
    The gripper links behind link_7 are lumped into link 7 (mass, centre of mass, inertia by the parallel-axis theorem)."""
    import xml.etree.ElementTree as ET
    model = dict(base or default())
    root = ET.parse(sdf_path).getroot()
    mdl = root.find(".//model")
    links, joints = {}, []

    def pose_of(node):
        txt = node.findtext("pose")
        v = np.array([float(x) for x in txt.split()]) if txt else np.zeros(6)
        T = np.eye(4)
        T[:3, :3], T[:3, 3] = _rpy_matrix(v[3:]), v[:3]
        return T

    for ln in mdl.findall("link"):
        inert = ln.find("inertial")
        I = np.zeros(3)
        mass, Tin = 0.0, np.eye(4)
        if inert is not None:
            mass = float(inert.findtext("mass"))
            Tin = pose_of(inert)
            I = np.array([float(inert.findtext("inertia/" + k)) for k in ("ixx", "iyy", "izz")])
        links[ln.get("name")] = {"T": pose_of(ln), "mass": mass, "Tin": Tin, "I": I}
    for jn in mdl.findall("joint"):
        joints.append({"name": jn.get("name"), "type": jn.get("type"), "parent": jn.findtext("parent"), "child": jn.findtext("child"),
                       "lower": jn.findtext("axis/limit/lower"), "upper": jn.findtext("axis/limit/upper"),
                       "damping": jn.findtext("axis/dynamics/damping")})
    chain = [j for j in joints if j["type"] == "revolute"][:7]
    assert len(chain) == 7, "expected 7 revolute arm joints"
    parent_T = links[chain[0]["parent"]]["T"]
    base_T = parent_T.copy()
    for i, j in enumerate(chain):
        child = links[j["child"]]
        rel = np.linalg.inv(parent_T) @ child["T"]
        # the library places the arm's base at Kuka.reset's base position itself (kuka.py:63): joint 0 is relative to the base link
        model["joint_xyz"][i], model["joint_rpy"][i] = rel[:3, 3], _matrix_rpy(rel[:3, :3])
        model["joint_lower"][i], model["joint_upper"][i] = float(j["lower"]), float(j["upper"])
        model["mass"][i], model["com"][i], model["inertia"][i] = child["mass"], child["Tin"][:3, 3], child["I"]
        parent_T = child["T"]
    if chain[0]["damping"] is not None:
        model["joint_damping"] = float(chain[0]["damping"])
    # lump everything behind link 7 (the gripper) rigidly into it
    tip = links[chain[-1]["child"]]
    behind, frontier = [], [chain[-1]["child"]]
    while frontier:
        cur = frontier.pop()
        for j in joints:
            if j["parent"] == cur and j["child"] not in behind and j not in chain:
                behind.append(j["child"])
                frontier.append(j["child"])
    m_tot = tip["mass"]
    c_tot = tip["mass"] * tip["Tin"][:3, 3]
    parts = [(tip["mass"], tip["Tin"][:3, 3], tip["Tin"][:3, :3] @ np.diag(tip["I"]) @ tip["Tin"][:3, :3].T)]
    for name in behind:
        ln = links[name]
        T = np.linalg.inv(tip["T"]) @ ln["T"] @ ln["Tin"]
        parts.append((ln["mass"], T[:3, 3], T[:3, :3] @ np.diag(ln["I"]) @ T[:3, :3].T))
        m_tot += ln["mass"]
        c_tot = c_tot + ln["mass"] * T[:3, 3]
    if m_tot > 0:
        c = c_tot / m_tot
        I = np.zeros((3, 3))
        for m, p, Ip in parts:
            d = p - c
            I += Ip + m * (np.dot(d, d) * np.eye(3) - np.outer(d, d))
        model["mass"][6], model["com"][6], model["inertia"][6] = m_tot, c, np.diag(I)      # off-diagonal terms dropped
    del base_T
    return model

# The tree lane-group kernel (csrc/kuka_tree.hpp) integrates the FULL model: 12 DoFs (arm + gripper_to_arm + two fingers + two
# tips), 12 motor rows.  Algorithmic float64 operations per env-step, free path (no limit / contact / friction row):
#   150 sweeps x (12 motor rows x (1 + 2 + 11 x 2) + 3 button rows x (1 + 2 + 2 x 2))                = 48 150
#   Gauss-Jordan inverse of the 12x12 mass matrix: 12 pivots x 12 rows x 12 columns x 2               =  3 460
#   CRBA: composite inertias over the tree, Ic S per link, 48 ancestor pairs x 6-term dot products    =  1 400
#   RNEA in world coordinates: per-link force ~190 x 12 (full inertia tensors), ancestor / descendant
#   sums 27 quantities x 12 lanes x ~5 terms x 2                                                      =  5 500
#   FK (12 frame compositions x 63 + 12 Rodrigues rotations x 45) + 12 sincos (~40 each)              =  1 780
#   IK on the arm block (as above)                                                                    =    990
#   row setup, 16 sphere-cylinder candidates, env logic, Philox                                       =    900
FLOPS_PER_ENV_STEP_TREE = 6.2e4


# ---------------------------------------------------------------------------------------------------------------------
# The full model (`srlhip_kuka_tree_model`, 510 doubles): the 12-DoF arm + gripper tree.
TREE_MODEL_DOUBLES = 510
# named offsets into the flat table (tests edit single entries)
TREE_JOINT0, TREE_JOINT_STRIDE, TREE_LOWER, TREE_UPPER = 1, 33, 16, 17
TREE_MAX_GENERIC_ROWS, TREE_FRICTION, TREE_SOLVER_DETAIL, TREE_CONTACT_ERP, TREE_LIMIT_ERP, TREE_LINEAR_SLOP = 504, 505, 506, 507, 508, 509
TREE_JOINT_FIELDS = (("parent", 1), ("xyz", 3), ("Rj", 9), ("axis", 3), ("lower", 1), ("upper", 1), ("damping", 1), ("mass", 1), ("com", 3),
                     ("inertia", 6), ("kp", 1), ("max_force", 1), ("max_vel", 1), ("joint_index", 1))          # 33 doubles per DoF


def tree_to_dict(table):
    """flat float64[510] -> {"nd", "joints": [dict x 12], "ee_link", "ee_point", "grip_link", "grip_point", "nsphere",
    "spheres": [dict(link, c, r, mu) x 16], "table_top_z", "button_base_z", "max_generic_rows", "friction", "solver_detail", "contact_erp",
    "limit_erp", "linear_slop"}"""
    t = np.asarray(table, dtype=np.float64).reshape(-1)
    assert t.shape == (TREE_MODEL_DOUBLES,)
    k = 1
    joints = []
    for _ in range(12):
        j = {}
        for name, n in TREE_JOINT_FIELDS:
            j[name] = t[k:k + n].copy() if n > 1 else float(t[k])
            k += n
        joints.append(j)
    out = {"nd": int(t[0]), "joints": joints, "ee_link": int(t[k]), "ee_point": t[k + 1:k + 4].copy(), "grip_link": int(t[k + 4]),
           "grip_point": t[k + 5:k + 8].copy(), "nsphere": int(t[k + 8])}
    k += 9
    out["spheres"] = [{"link": int(t[k + 6 * i]), "c": t[k + 6 * i + 1:k + 6 * i + 4].copy(), "r": float(t[k + 6 * i + 4]), "mu": float(t[k + 6 * i + 5])} for i in range(16)]
    k += 96
    out["table_top_z"], out["button_base_z"], out["max_generic_rows"], out["friction"] = float(t[k]), float(t[k + 1]), int(t[k + 2]), int(t[k + 3])
    out["solver_detail"], out["contact_erp"], out["limit_erp"], out["linear_slop"] = int(t[k + 4]), float(t[k + 5]), float(t[k + 6]), float(t[k + 7])
    return out


def tree_to_table(m):
    t = [float(m["nd"])]
    for j in m["joints"]:
        for name, n in TREE_JOINT_FIELDS:
            t.extend(np.asarray(j[name], dtype=np.float64).reshape(-1).tolist())
    t.append(float(m["ee_link"])); t.extend(np.asarray(m["ee_point"], dtype=np.float64).tolist())
    t.append(float(m["grip_link"])); t.extend(np.asarray(m["grip_point"], dtype=np.float64).tolist())
    t.append(float(m["nsphere"]))
    for s in m["spheres"]:
        t.append(float(s["link"])); t.extend(np.asarray(s["c"], dtype=np.float64).tolist()); t.extend([float(s["r"]), float(s["mu"])])
    t.extend([float(m["table_top_z"]), float(m["button_base_z"]), float(m["max_generic_rows"]), float(m["friction"])])
    t.extend([float(m["solver_detail"]), float(m["contact_erp"]), float(m["limit_erp"]), float(m["linear_slop"])])
    t = np.asarray(t, dtype=np.float64)
    assert t.shape == (TREE_MODEL_DOUBLES,)
    return t


def tree_default():
    from . import _lib
    return tree_to_dict(_lib.kuka_tree_default_model())


def _quat_matrix(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def tree_from_pybullet(p, uid, base=None):
    """The full-model table read from a LOADED body through PyBullet's own introspection (p = the pybullet module, uid = the body
    kuka.py:60 loads): joint tree, joint frames, axes, limits, damping from getJointInfo; link masses, inertial frames and
    inertias from getDynamicsInfo; link frames at q = 0 from getLinkState.  Fixed joints (9 and 12 of kuka_with_gripper2.sdf) are
    merged into their parent bodies exactly.  What PyBullet does not carry per joint — the motors the reference commands every
    step (kuka.py:167-187) — and the collision spheres come from `base` (default: the baked table); sphere friction is
    recomputed from the links' lateral friction (x 0.5 for the button / table)."""
    base = base or tree_default()
    nj = p.getNumJoints(uid)
    saved = [p.getJointState(uid, j)[0] for j in range(nj)]
    for j in range(nj):
        p.resetJointState(uid, j, 0.0)
    try:
        bpos, born = p.getBasePositionAndOrientation(uid)
        Tb = np.eye(4); Tb[:3, :3] = _quat_matrix(born); Tb[:3, 3] = bpos
        info = [p.getJointInfo(uid, j) for j in range(nj)]
        frames, inert = {-1: Tb}, {}
        for j in range(nj):
            ls = p.getLinkState(uid, j, computeForwardKinematics=True)
            T = np.eye(4); T[:3, :3] = _quat_matrix(ls[5]); T[:3, 3] = ls[4]                 # URDF link frame in the world
            frames[j] = T
            dyn = p.getDynamicsInfo(uid, j)
            Rin = _quat_matrix(dyn[4])
            inert[j] = {"mass": dyn[0], "c": np.array(dyn[3]), "I": Rin @ np.diag(dyn[2]) @ Rin.T, "mu": dyn[1]}
        movable = [j for j in range(nj) if info[j][2] != p.JOINT_FIXED]
        assert len(movable) == 12, "expected 12 movable joints, found {}".format(len(movable))
        dof_of = {j: i for i, j in enumerate(movable)}

        def body_of(link):                       # the DoF whose body a link belongs to (fixed links hang on their parent's)
            while link not in dof_of:
                link = info[link][16]
            return dof_of[link]

        m = {k: v for k, v in base.items()}
        m["joints"] = [dict(j) for j in base["joints"]]
        parts = {i: [] for i in range(12)}
        for j in range(nj):
            i = body_of(j)
            Trel = np.linalg.inv(frames[movable[i]]) @ frames[j]                              # identity for the DoF's own link
            parts[i].append((inert[j]["mass"], Trel[:3, :3] @ inert[j]["c"] + Trel[:3, 3], Trel[:3, :3] @ inert[j]["I"] @ Trel[:3, :3].T))
        for i, j in enumerate(movable):
            J = m["joints"][i]
            par = info[j][16]
            Tpar = frames[movable[body_of(par)]] if par >= 0 else Tb
            rel = np.linalg.inv(Tpar) @ frames[j]
            J["parent"] = float(body_of(par)) if par >= 0 else -1.0
            # the library places the arm's base at Kuka.reset's base position itself (kuka.py:63): DoF 0 is relative to the base link
            J["xyz"], J["Rj"] = rel[:3, 3].copy(), rel[:3, :3].reshape(-1).copy()
            J["axis"] = np.array(info[j][13], dtype=np.float64)
            J["lower"], J["upper"] = float(info[j][8]), float(info[j][9])
            if abs(J["lower"]) > 9.0 and abs(J["upper"]) > 9.0 or info[j][8] > info[j][9]:     # +-10 rad "limits" of the gripper joints never act
                J["lower"], J["upper"] = 1.0, -1.0
            J["damping"] = float(info[j][6])
            mt = sum(q[0] for q in parts[i])
            c = sum(q[0] * q[1] for q in parts[i]) / mt
            I = np.zeros((3, 3))
            for mass, ci, Ii in parts[i]:
                r = ci - c
                I += Ii + mass * (np.dot(r, r) * np.eye(3) - np.outer(r, r))
            J["mass"], J["com"] = float(mt), c
            J["inertia"] = np.array([I[0, 0], I[0, 1], I[0, 2], I[1, 1], I[1, 2], I[2, 2]])
            J["joint_index"] = float(j)
        ee = movable[int(base["ee_link"])]
        m["ee_point"] = inert[ee]["c"].copy()                                                 # IK end effector = the link's inertial frame
        gl = movable[int(base["grip_link"])]
        m["grip_point"] = inert[gl]["c"].copy()                                               # getLinkState(.., 8)[0]: COM of that link alone
        for s in m["spheres"][:int(m["nsphere"])]:
            s["mu"] = float(inert[movable[int(s["link"])]]["mu"]) * 0.5
        return m
    finally:
        for j in range(nj):
            p.resetJointState(uid, j, saved[j])
