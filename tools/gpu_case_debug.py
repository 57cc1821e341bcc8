"""Debug helper: MGBHIP_DEBUG=1 python tools/gpu_case_debug.py L p  -> Newton trace of fem2d_P2 on the default hierarchy."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mgb_amd as m
from mgb_amd.solve import MGBConvergenceFailure
L = int(sys.argv[1]); p = float(sys.argv[2])
prob = m.assemble(m.amg(m.subdivide(m.fem2d_P2(), L)), p=p)
t = time.time()
try:
    sol = m.mgb_solve(prob)
    print("converged: its", int(np.sum(sol.SOL_main["its"])), time.time() - t)
except MGBConvergenceFailure as e:
    print("failure:", e.code, str(e)[:160], time.time() - t)
