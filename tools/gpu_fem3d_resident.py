import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import mgb_amd as m
from mgb_amd.solve import mgb_driver
prob = m.assemble(m.amg(m.subdivide(m.fem3d(k=1), 6), prolongator=m.amg_ruge_stuben(max_coarse=300)), p=4.0)
t0 = time.time(); sol = m.mgb_solve(prob, keep_device=True); print("first solve wall", round(time.time()-t0, 2), flush=True)
t0 = time.time(); S = mgb_driver(sol.device); dt = time.time()-t0
its = int(S["SOL_main"]["its"].sum()) + int(S["SOL_feasibility"]["its"].sum())
print("second solve (plans + symbolic analysis resident) wall", round(dt, 2), "its", its, "it/s", round(its/dt, 1), flush=True)
print("feas its per level", S["SOL_feasibility"]["its"].sum(axis=1).tolist(), "main", S["SOL_main"]["its"].sum(axis=1).tolist())
sol.device.close()
