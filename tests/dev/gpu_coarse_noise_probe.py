"""Where does the noise in lambda^2 come from on a tiny creeping level?  (development only; imports the oracle)

    python tests/dev/gpu_coarse_noise_probe.py [L=8] [p=1.5] [level=0] [maxit=80]

Replays the Newton solve of the coarsest level of the reference-default ladder (the one the initial centring ends
in, src/mgb.jl:64-73) in Python on the DEVICE closures, and at every iterate compares
  * the device solve of the device's H, g with LAPACK and with an 80-bit elimination of the same H, g  (solver noise)
  * the device's H, g with the oracle's at the same point                                               (assembly noise)
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import mgb_amd as m
from mgb_amd.device import DeviceMGBProblem
from oracle import mgb_oracle as O

L = int(sys.argv[1]) if len(sys.argv) > 1 else 8
p = float(sys.argv[2]) if len(sys.argv) > 2 else 1.5
lev = int(sys.argv[3]) if len(sys.argv) > 3 else 0
maxit = int(sys.argv[4]) if len(sys.argv) > 4 else 80
with_oracle = os.environ.get("NO_ORACLE", "0") != "1"
prob = m.assemble(m.amg(m.subdivide(m.fem2d_P2(), L)), p=p)
print("ladder", [R.shape[1] for R in prob.M[0].R_fine], flush=True)
D = DeviceMGBProblem(prob)
main = D.main
Mo = O.OracleAMG(prob.M[0])
B = O.Barrier(prob.Q)
R = Mo.R_fine[lev]
z0 = np.ascontiguousarray(prob.g.T).reshape(-1).copy()
c = 0.1 * prob.f
msz = R.shape[1]


def ld_solve(H, g):
    """Gaussian elimination with partial pivoting in 80-bit arithmetic."""
    A = np.array(H, dtype=np.longdouble)
    b = np.array(g, dtype=np.longdouble)
    n = b.size
    for k in range(n):
        pv = k + int(np.argmax(np.abs(A[k:, k])))
        if pv != k:
            A[[k, pv]] = A[[pv, k]]
            b[[k, pv]] = b[[pv, k]]
        for i in range(k + 1, n):
            f = A[i, k] / A[k, k]
            A[i, k:] -= f * A[k, k:]
            b[i] -= f * b[k]
    x = np.zeros(n, dtype=np.longdouble)
    for k in range(n - 1, -1, -1):
        x[k] = (b[k] - np.dot(A[k, k + 1:], x[k + 1:])) / A[k, k]
    return x


s = np.zeros(msz)
y = main.f0(lev, s, c, z0)
g = main.f1(lev, s, c, z0)
print(f"{'k':>4} {'y':>22} {'lam2_dev':>12} {'lam2_lapack':>12} {'lam2_80bit':>12} {'cond(H)':>9} "
      f"{'dH/|H|':>9} {'dg/|g|':>9} {'lam2_orc80':>12} {'step':>8}", flush=True)
for k in range(1, maxit + 1):
    Hd = main.f2(lev, s, c, z0).toarray()
    n_dev = main.solve(lev, g)
    lam_dev = float(g @ n_dev)
    lam_np = float(g @ np.linalg.solve(Hd, g))
    x80 = ld_solve(Hd, g)
    lam_80 = float(np.dot(np.array(g, dtype=np.longdouble), x80))
    ev = np.linalg.eigvalsh(0.5 * (Hd + Hd.T))
    cond = ev[-1] / ev[0] if ev[0] > 0 else float("inf")
    dH = dg = lam_o = float("nan")
    if with_oracle:
        Ho = np.asarray(B.f2(s, Mo.w, c, R, Mo.D_fine, z0).todense())
        go = B.f1(s, Mo.w, c, R, Mo.D_fine, z0)
        dH = float(np.abs(Hd - Ho).max() / np.abs(Ho).max())
        dg = float(np.abs(g - go).max() / np.abs(go).max())
        lam_o = float(np.dot(np.array(go, dtype=np.longdouble), ld_solve(Ho, go)))
    inc = lam_dev
    # backtracking line search on the device closures (src/newton.jl:139-154)
    st = 1.0
    if inc <= 0:
        print(f"{k:4d} {y:22.15e} {lam_dev:12.4e} {lam_np:12.4e} {lam_80:12.4e} {cond:9.2e} {dH:9.2e} {dg:9.2e} {lam_o:12.4e}   (inc<=0: stop)")
        break
    while st > 0:
        sn = s - st * n_dev
        yn = main.f0(lev, sn, c, z0)
        gn = main.f1(lev, sn, c, z0)
        if np.isfinite(yn) and np.all(np.isfinite(gn)) and (np.array_equal(sn, s) or yn <= y - 0.1 * inc * st):
            break
        st *= 0.5
    print(f"{k:4d} {y:22.15e} {lam_dev:12.4e} {lam_np:12.4e} {lam_80:12.4e} {cond:9.2e} {dH:9.2e} {dg:9.2e} {lam_o:12.4e} {st:8.2e}", flush=True)
    nn = prob.M[0].w.size
    if np.sqrt(inc) < 0.25 / np.sqrt(nn) or (yn >= y and np.linalg.norm(gn) >= 0.9 * np.linalg.norm(g)):
        print("stopping criterion met (approx: ymin/gmin not tracked)")
        s, y, g = sn, yn, gn
        break
    s, y, g = sn, yn, gn
D.close()
