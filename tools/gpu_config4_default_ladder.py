import sys, os, time
sys.path.insert(0, "/root/repo")
import numpy as np, mgb_amd as m
prob = m.assemble(m.amg(m.subdivide(m.fem3d(k=1), 6)), p=4.0)
print("levels", [R.shape[1] for R in prob.M[0].R_fine], "phase-I levels", [R.shape[1] for R in prob.M[1].R_fine], flush=True)
t0 = time.time()
try:
    sol = m.mgb_solve(prob)
    print("converged", sol.SOL_main["its"].sum(), "feas", None if sol.SOL_feasibility is None else sol.SOL_feasibility["its"].sum(axis=1).tolist())
except Exception as e:
    print("FAILED:", type(e).__name__, str(e)[:600])
print("wall", time.time() - t0)
