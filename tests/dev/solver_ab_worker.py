"""Worker of tests/test_gpu_solver.py (one process per kernel selection: the library reads its A/B switches once).

    python tests/dev/solver_ab_worker.py OUT.npz

Graded SPD matrices on every level's Hessian pattern are pushed through mgbhip_set_hessian and solved on both solve
paths (generic forward + backward sweeps, and the bordered factorization the Newton loop uses).  Solutions go to
OUT.npz; the parent compares kernel selections and measures backward errors."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from mgb_amd.device import DeviceMGBProblem
from solver_cases import CASES, build, key_of, matrices

out = {}
for fam, L, p, rs in CASES:
    D = DeviceMGBProblem(build(fam, L, p, rs))
    P = D.main
    for lev, grade, A, g in matrices(P):
        P.set_hessian(lev, A.data)
        x1 = P.solve(lev, g)
        P.set_hessian(lev, A.data)
        x2, lam, status = P.solve_newton(lev, g, check=False)
        key = key_of(fam, L, p, rs, lev, grade)
        out[key + "_x"], out[key + "_xn"], out[key + "_lam"] = x1, x2, np.array([lam, float(status)])
    D.close()
np.savez(sys.argv[1], **out)
print("worker done:", len(out), "arrays")
