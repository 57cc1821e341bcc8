"""The oracle's Newton solve in the COARSEST space of the reference-default ladder from the default start -- the solve
the initial centring ends in after the finer attempts have failed without moving z (src/mgb.jl:10-15, :64-73:
maxit = 10000 there).  Development only.   python tests/dev/oracle_coarsest_newton.py L p [level=0]"""
import os, sys, time, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import mgb_amd as m
from oracle import mgb_oracle as O
L, p = int(sys.argv[1]), float(sys.argv[2])
lev = int(sys.argv[3]) if len(sys.argv) > 3 else 0
prob = m.assemble(m.amg(m.subdivide(m.fem2d_P2(), L)), p=p)
M = O.OracleAMG(prob.M[0])
print("ladder", [R.shape[1] for R in M.R_fine], "level", lev, flush=True)
B = O.Barrier(prob.Q)
R = M.R_fine[lev]
z = np.ascontiguousarray(prob.g.T).reshape(-1).copy()
c = 0.1 * prob.f
n = M.w.size
t0 = time.time()


def trace(k, y, inc, H, g, nn):
    if k % 25 == 0 or inc <= 0 or k <= 3:
        Hd = H.toarray() if hasattr(H, "toarray") else np.asarray(H)
        ev = np.linalg.eigvalsh(0.5 * (Hd + Hd.T))
        print(f"[{time.time()-t0:7.0f}s] k={k:5d} y={y:.15e} lambda2={inc:.6e} |g|={np.linalg.norm(g):.3e} eig[{ev[0]:.3e},{ev[-1]:.3e}]", flush=True)


SOL = O.newton(lambda s: B.f0(s, M.w, c, R, M.D_fine, z), lambda s: B.f1(s, M.w, c, R, M.D_fine, z),
               lambda s: B.f2(s, M.w, c, R, M.D_fine, z), np.zeros(R.shape[1]), maxit=10000,
               stopping_criterion=O.stopping_inexact(0.25 / math.sqrt(n), 0.9), line_search=O.linesearch_backtracking(),
               stats={"trace": trace})
print("RESULT converged", SOL["converged"], "k", SOL["k"], "y", SOL["y"], f"{time.time()-t0:.0f}s", flush=True)
