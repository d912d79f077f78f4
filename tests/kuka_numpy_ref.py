"""Independent numpy formulation of the Kuka arm dynamics used to cross-check the
C oracle's Featherstone ABA (tests only).  Lagrangian/Jacobian form:
    M(q) = sum_i Jv_i^T m_i Jv_i + Jw_i^T (R_i Ic_i R_i^T) Jw_i          (COM Jacobians)
    G(q) = -sum_i Jv_i^T m_i g
    h(q, qd) = sum_i Jv_i^T m_i (dJv_i qd) + Jw_i^T (I_i dJw_i qd + w_i x I_i w_i)
with dJ*qd obtained by central differences of J along qd.  Shares only the model
constants (re-typed here from SURVEY.md App. B.4) with the oracle."""
import numpy as np

PI = np.pi
BASE = np.array([-0.1, 0.0, -0.15])
XYZ = np.array([[0, 0, 0.1575], [0, 0, 0.2025], [0, 0.2045, 0], [0, 0, 0.2155], [0, 0.1845, 0], [0, 0, 0.2155],
                [0, 0.081, 0]])
RPY = np.array([[0, 0, 0], [PI / 2, 0, PI], [PI / 2, 0, PI], [PI / 2, 0, 0], [-PI / 2, PI, 0], [PI / 2, 0, 0],
                [-PI / 2, PI, 0]])
MASS = np.array([4.0, 4.0, 3.0, 2.7, 1.7, 1.8, 1.8])
COM = np.array([[0, -0.03, 0.12], [0.0003, 0.059, 0.042], [0, 0.03, 0.13], [0, 0.067, 0.034], [0.0001, 0.021, 0.076],
                [0, 0.0006, 0.0004], [0, 0, 0.31 / 3.0]])
INERTIA = np.array([[0.1, 0.09, 0.02], [0.05, 0.018, 0.044], [0.08, 0.075, 0.01], [0.03, 0.01, 0.029],
                    [0.02, 0.018, 0.005], [0.005, 0.0036, 0.0047], [0.0075, 0.0075, 0.003]])


def rot(axis, a):
    c, s = np.cos(a), np.sin(a)
    if axis == "x":
        return np.array([[1, 0, 0], [0, c, -s], [0, s, c]])
    if axis == "y":
        return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])


def fk(q):
    R, p = np.eye(3), BASE.copy()
    Rs, ps = [], []
    for i in range(7):
        p = p + R @ XYZ[i]
        R = R @ rot("z", RPY[i, 2]) @ rot("y", RPY[i, 1]) @ rot("x", RPY[i, 0]) @ rot("z", q[i])
        Rs.append(R)
        ps.append(p)
    return np.array(Rs), np.array(ps)


def com_jacobians(q):
    Rs, ps = fk(q)
    out = []
    for i in range(7):
        c = ps[i] + Rs[i] @ COM[i]
        Jv, Jw = np.zeros((3, 7)), np.zeros((3, 7))
        for j in range(i + 1):
            z = Rs[j][:, 2]
            Jv[:, j] = np.cross(z, c - ps[j])
            Jw[:, j] = z
        out.append((Jv, Jw, Rs[i] @ np.diag(INERTIA[i]) @ Rs[i].T))
    return out


def mass_matrix(q):
    M = np.zeros((7, 7))
    for i, (Jv, Jw, I) in enumerate(com_jacobians(q)):
        M += MASS[i] * Jv.T @ Jv + Jw.T @ I @ Jw
    return M


def bias(q, qd, gz=-10.0, eps=1e-6):
    g = np.array([0, 0, gz])
    J0 = com_jacobians(q)
    Jp = com_jacobians(q + eps * qd)
    Jm = com_jacobians(q - eps * qd)
    h = np.zeros(7)
    for i in range(7):
        Jv, Jw, I = J0[i]
        dJv = (Jp[i][0] - Jm[i][0]) / (2 * eps)
        dJw = (Jp[i][1] - Jm[i][1]) / (2 * eps)
        w = Jw @ qd
        h += Jv.T @ (MASS[i] * (dJv @ qd)) + Jw.T @ (I @ (dJw @ qd) + np.cross(w, I @ w))
        h -= Jv.T @ (MASS[i] * g)
    return h


def forward_dynamics(q, qd, tau, gz=-10.0):
    return np.linalg.solve(mass_matrix(q), tau - bias(q, qd, gz))
