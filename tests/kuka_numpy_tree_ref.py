"""Independent numpy formulation of the FULL Kuka model's dynamics (the 12-DoF arm + gripper tree) used to cross-check the C
oracle's tree ABA (tests only).  Everything is derived from the 506-double model table alone — parents, joint frames, axes,
masses, centres of mass, full inertia tensors — in Lagrangian / Jacobian form:
    M(q) = sum_i Jv_i^T m_i Jv_i + Jw_i^T (R_i Ic_i R_i^T) Jw_i          (COM Jacobians over each link's ancestors)
    h(q, qd) = sum_i Jv_i^T m_i (dJv_i qd - g) + Jw_i^T (I_i dJw_i qd + w_i x I_i w_i)
with dJ qd by central differences.  Shares no code with oracle/kuka_oracle.c."""
import numpy as np

BASE = np.array([-0.1, 0.0, -0.15])          # kuka.py:63


def unpack(table):
    t = np.asarray(table, dtype=np.float64)
    J = []
    for i in range(12):
        r = t[1 + 33 * i:1 + 33 * (i + 1)]
        I6 = r[23:29]
        J.append({"parent": int(r[0]), "xyz": r[1:4], "Rj": r[4:13].reshape(3, 3), "axis": r[13:16], "mass": r[19], "com": r[20:23],
                  "I": np.array([[I6[0], I6[1], I6[2]], [I6[1], I6[3], I6[4]], [I6[2], I6[4], I6[5]]])})
    return J


def axis_rot(a, q):
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(q) * K + (1 - np.cos(q)) * (K @ K)


def fk(J, q):
    R, p = [None] * 12, [None] * 12
    for i, j in enumerate(J):
        Rp, pp = (np.eye(3), BASE) if j["parent"] < 0 else (R[j["parent"]], p[j["parent"]])
        p[i] = pp + Rp @ j["xyz"]
        R[i] = Rp @ j["Rj"] @ axis_rot(j["axis"], q[i])
    return np.array(R), np.array(p)


def ancestors(J, i):
    out = []
    while i >= 0:
        out.append(i)
        i = J[i]["parent"]
    return out


def com_jacobians(J, q):
    R, p = fk(J, q)
    out = []
    for i, j in enumerate(J):
        c = p[i] + R[i] @ j["com"]
        Jv, Jw = np.zeros((3, 12)), np.zeros((3, 12))
        for k in ancestors(J, i):
            z = R[k] @ J[k]["axis"]
            Jv[:, k] = np.cross(z, c - p[k])
            Jw[:, k] = z
        out.append((Jv, Jw, R[i] @ j["I"] @ R[i].T))
    return out


def mass_matrix(J, q):
    M = np.zeros((12, 12))
    for j, (Jv, Jw, I) in zip(J, com_jacobians(J, q)):
        M += j["mass"] * Jv.T @ Jv + Jw.T @ I @ Jw
    return M


def bias(J, q, qd, gz=-10.0, eps=1e-6):
    g = np.array([0, 0, gz])
    J0, Jp, Jm = com_jacobians(J, q), com_jacobians(J, q + eps * qd), com_jacobians(J, q - eps * qd)
    h = np.zeros(12)
    for i, j in enumerate(J):
        Jv, Jw, I = J0[i]
        dJv, dJw = (Jp[i][0] - Jm[i][0]) / (2 * eps), (Jp[i][1] - Jm[i][1]) / (2 * eps)
        w = Jw @ qd
        h += Jv.T @ (j["mass"] * (dJv @ qd - g)) + Jw.T @ (I @ (dJw @ qd) + np.cross(w, I @ w))
    return h


def forward_dynamics(J, q, qd, tau, gz=-10.0):
    return np.linalg.solve(mass_matrix(J, q), tau - bias(J, q, qd, gz))
