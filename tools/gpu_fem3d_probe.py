"""config 4 probe: default start (phase I) with the reference-default hierarchy and with max_coarse=300."""
import sys, os, time, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import mgb_amd as m
from mgb_amd.solve import mgb_driver
for kw in ({}, {"max_coarse": 300}):
    t0 = time.time()
    prob = m.assemble(m.amg(m.subdivide(m.fem3d(k=1), 6), prolongator=m.amg_ruge_stuben(**kw)), p=4.0)
    print(kw, "setup %.1f" % (time.time() - t0), "levels", [R.shape[1] for R in prob.M[0].R_fine], flush=True)
    try:
        t0 = time.time(); sol = m.mgb_solve(prob, keep_device=True); print("  first solve wall", round(time.time() - t0, 2), flush=True)
        t0 = time.time(); S = mgb_driver(sol.device); dt = time.time() - t0
        its = int(S["SOL_main"]["its"].sum()) + int(S["SOL_feasibility"]["its"].sum())
        print("  resident solve wall", round(dt, 2), "its", its, "it/s", round(its / dt, 1), flush=True)
        print("  feas its per level", S["SOL_feasibility"]["its"].sum(axis=1).tolist(), "main", S["SOL_main"]["its"].sum(axis=1).tolist(), flush=True)
        print("  bitwise equal:", np.array_equal(S["z"], sol.z))
        sol.device.close()
    except Exception as e:
        print("  FAILED", str(e)[:300], round(time.time() - t0, 1), flush=True)
