"""Round 5 lead, NOT built (profiles/NOTES.md section S): could the 150 Gauss-Seidel sweeps of a contact-free Kuka step be replaced by K
sequential sweeps + a closed-form jump (matrix powers of the affine sweep operator on the rows whose clamp status has settled)?
Dumps the 15-row problem of every contact-free step from an INSTRUMENTED COPY of the oracle under /tmp (never the shipped one), replays
the sweeps in numpy (reconstruction check), and reports for K = 8 / 16 / 24 / 32: how often the status is final after K sweeps, the
error of the jump against the sequential result, the spectral radius of the sweep operator, whether a norm-bound certificate could
prove that no status flips later, and whether late flips are visible in the final status.  Usage (CPU, repo root):
    python profiles/probes/pgs_closed_form_probe.py"""
import ctypes, os, subprocess, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "robotics-rl-srl_amd"))
import numpy as np
src = open("oracle/kuka_oracle.c").read()
a = "    for (it = 0; it < KM_SOLVER_ITERS; it++) {\n        for (jj = 0; jj < nrows; jj++) {"
assert a in src
src = src.replace(a, """    if (nrows == 15 && g_dump) {
        double rec[15*15 + 15*3]; int jj2, kk2, ii2;
        for (jj2 = 0; jj2 < 15; jj2++) for (kk2 = 0; kk2 < 15; kk2++) {
            double s2 = (rows[jj2].bsel == rows[kk2].bsel ? rows[jj2].Jb * rows[kk2].WJb : 0.0);
            for (ii2 = 0; ii2 < n; ii2++) s2 += rows[jj2].J[ii2] * rows[kk2].WJ[ii2];
            rec[jj2*15+kk2] = s2 * rows[jj2].Dinv;
        }
        for (jj2 = 0; jj2 < 15; jj2++) { rec[225+jj2] = rows[jj2].rhs; rec[240+jj2] = rows[jj2].lo; rec[255+jj2] = rows[jj2].hi; }
        fwrite(rec, sizeof rec, 1, g_dump);
    }
""" + a)
b = "    /* -- semi-implicit Euler -- */\n    for (i = 0; i < n; i++) { e->qd[i] += dv[i]; e->q[i] += dt * e->qd[i]; }"
assert b in src
src = src.replace(b, """    if (nrows == 15 && g_dump) { double fin[15]; int jj3; for (jj3 = 0; jj3 < 15; jj3++) fin[jj3] = rows[jj3].applied; fwrite(fin, sizeof fin, 1, g_dump); }
""" + b)
src = src.replace("#include <math.h>", "#include <math.h>\n#include <stdio.h>\nstatic FILE *g_dump = NULL;\nvoid pgs_dump_open(const char *p) { g_dump = fopen(p, \"wb\"); }\nvoid pgs_dump_close(void) { if (g_dump) fclose(g_dump); g_dump = NULL; }", 1)
open("/tmp/orc_dump/kuka_oracle.c", "w").write(src)
for f in os.listdir("oracle"):
    if f.endswith(".h") or (f.endswith(".c") and f != "kuka_oracle.c"):
        open("/tmp/orc_dump/" + f, "w").write(open("oracle/" + f).read())
subprocess.check_call("cd /tmp/orc_dump && gcc -O2 -std=c99 -fPIC -ffp-contract=off -w -shared -o liboracle_dump.so *.c -lm", shell=True)
import oracle.clib as C
C.lib()
C._lib = ctypes.CDLL("/tmp/orc_dump/liboracle_dump.so")
from oracle import kuka_clib
kuka_clib.set_full(True)
C._lib.pgs_dump_open(b"/tmp/orc_dump/pgs.bin")
out = kuka_clib.rollout(np.arange(12), 700, actions=None, rng_mode=kuka_clib.RNG_PHILOX, trace=False)
C._lib.pgs_dump_close()
print("episodes finished:", int(out["done"].sum()), "file MB", os.path.getsize("/tmp/orc_dump/pgs.bin") / 1e6)

# ---- analysis
raw = np.fromfile("/tmp/orc_dump/pgs.bin", dtype=np.float64)
rec = 270 + 15
n = len(raw) // rec
raw = raw[:n * rec].reshape(n, rec)
A = raw[:, :225].reshape(n, 15, 15); rhs = raw[:, 225:240]; lo = raw[:, 240:255]; hi = raw[:, 255:270]; fin = raw[:, 270:]
print("steps", n)
def sweep(A, rhs, lo, hi, lam, iters):
    # oracle: lam_k <- clamp(lam_k + rhs_k - (J_k dv) Dinv_k), A[k][j] = J_k.WJ_j * Dinv_k, dv = sum_j lam_j WJ_j
    lam = lam.copy(); hist = []
    for it in range(iters):
        for k in range(15):
            s = lam[k] + rhs[k] - A[k] @ lam
            lam[k] = min(max(s, lo[k]), hi[k])
        hist.append(lam.copy())
    return lam, hist
# check reconstruction on a few steps
idx = np.random.RandomState(0).choice(n, 300, replace=False)
err = 0
for i in idx[:20]:
    l, _ = sweep(A[i], rhs[i], lo[i], hi[i], np.zeros(15), 150)
    err = max(err, np.abs(l - fin[i]).max() / max(1e-30, np.abs(fin[i]).max()))
print("reconstruction rel err", err)
def analyze(K):
    res = {"cert": 0, "noflip_after_K": 0, "jump_err": [], "rho": [], "n": 0, "free_rows": []}
    for i in idx:
        a, r, l, h = A[i], rhs[i], lo[i], hi[i]
        lamK, hist = sweep(a, r, l, h, np.zeros(15), 150)
        st = lambda x: np.where(x <= l, -1, np.where(x >= h, 1, 0))
        sK = st(hist[K - 1])
        stable = all((st(hist[t]) == sK).all() for t in range(K, 150))
        res["noflip_after_K"] += stable
        res["n"] += 1
        free = np.nonzero(sK == 0)[0]
        res["free_rows"].append(len(free))
        # linear GS on the free rows with the clamped ones fixed: x <- G x + g (one full sweep)
        x0 = hist[K - 1].copy()
        def one(x):
            x = x.copy()
            for k in range(15):
                if sK[k] == 0: x[k] = x[k] + r[k] - a[k] @ x
            return x
        m = len(free)
        if m == 0:
            res["cert"] += 1; res["jump_err"].append(0.0); continue
        base = x0.copy(); base[free] = 0.0
        g = one(base)[free]
        G = np.zeros((m, m))
        for c, f in enumerate(free):
            e = base.copy(); e[f] = 1.0
            G[:, c] = one(e)[free] - g
        rho = np.abs(np.linalg.eigvals(G)).max()
        res["rho"].append(rho)
        # jump 150-K sweeps by affine powers
        M = np.eye(m + 1); M[:m, :m] = G; M[:m, m] = g
        P = np.linalg.matrix_power(M, 150 - K)
        xj = x0.copy(); xj[free] = P[:m, :m] @ x0[free] + P[:m, m]
        if stable: res["jump_err"].append(np.abs(xj - hist[-1]).max() / max(1e-30, np.abs(hist[-1]).max()))
        # certificate: margins at sweep K vs bound on all future movement, using norm of G^8 (inf norm)
        d = one(x0)[free] - x0[free]
        G8 = np.linalg.matrix_power(G, 8); beta = np.abs(G8).sum(axis=1).max()
        # movement over one block of 8 sweeps bounded by sum_{r<8} |G^r d|
        mv = np.zeros(m); v = d.copy()
        for rr in range(8): mv += np.abs(v); v = G @ v
        if beta < 1:
            bound = mv.max() / (1 - beta)          # crude: total future displacement of any free row
            # free rows: distance to bounds; clamped rows: how far the unclamped value is beyond the bound, moves by at most |a_kfree| . bound
            ok = True
            for k in range(15):
                if sK[k] == 0:
                    ok &= (x0[k] - l[k] > bound) and (h[k] - x0[k] > bound)
                else:
                    s = x0[k] + r[k] - a[k] @ x0
                    slack = (l[k] - s) if sK[k] < 0 else (s - h[k])
                    ok &= slack > np.abs(a[k][free]).sum() * bound
            res["cert"] += ok
    return res
for K in (8, 16, 24, 32):
    r = analyze(K)
    je = np.array(r["jump_err"])
    print("K=%d: no flip after K in %.1f%% of steps; certificate passes %.1f%%; jump rel err max %.2e; rho max %.3f median %.3f; free rows median %d" % (
        K, 100.0 * r["noflip_after_K"] / r["n"], 100.0 * r["cert"] / r["n"], je.max() if len(je) else -1, max(r["rho"]), np.median(r["rho"]), np.median(r["free_rows"])))
print("---- details")
K = 24
undetected = 0; late = 0; gaps = []
for i in idx:
    a, r, l, h = A[i], rhs[i], lo[i], hi[i]
    lamK, hist = sweep(a, r, l, h, np.zeros(15), 150)
    st = lambda x: np.where(x <= l, -1, np.where(x >= h, 1, 0))
    sK = st(hist[K - 1])
    flips = [t for t in range(K, 150) if not (st(hist[t]) == sK).all()]
    if flips:
        late += 1
        if (st(hist[-1]) == sK).all(): undetected += 1
    free = np.nonzero(sK == 0)[0]
    Af = a[np.ix_(free, free)]
    # GS iteration matrix of the free block: (D + L)^-1 (-U) with D = I (the update is x_k += r_k - a_k.x: a_kk = 1 after Dinv scaling?)
    Lw = np.tril(Af); U = np.triu(Af, 1)
    G = -np.linalg.solve(Lw, U)
    ev = np.abs(np.linalg.eigvals(G)); gaps.append(1 - ev.max())
print("K=24: steps with a flip after K: %d of %d; of those NOT visible in the status at sweep 150: %d" % (late, len(idx), undetected))
gaps = np.array(gaps)
print("1 - rho(G): min %.2e median %.2e max %.2e  -> rho^126 median %.3f" % (gaps.min(), np.median(gaps), gaps.max(), (1 - np.median(gaps)) ** 126))
