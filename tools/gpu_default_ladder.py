"""fem2d_P2 on the reference-default ladder amg(subdivide(fem2d_P2(), L)): iteration counts per level and t-step.
python tools/gpu_default_ladder.py L p"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mgb_amd as m
from mgb_amd.solve import MGBConvergenceFailure
L, p = int(sys.argv[1]), float(sys.argv[2])
prob = m.assemble(m.amg(m.subdivide(m.fem2d_P2(), L)), p=p)
print("ladder", [R.shape[1] for R in prob.M[0].R_fine], flush=True)
t = time.time()
try:
    sol = m.mgb_solve(prob)
    its = sol.SOL_main["its"]
    print(json.dumps(dict(L=L, p=p, converged=True, seconds=time.time() - t, total=int(its.sum()), per_level=its.sum(axis=1).tolist(),
                          first_step_per_level=its[:, 0].tolist(), t_steps=int(its.shape[1]))))
except MGBConvergenceFailure as e:
    print(json.dumps(dict(L=L, p=p, converged=False, code=e.code, message=str(e)[:200], seconds=time.time() - t)))
