"""Development sweep: device vs oracle `mgb_solve` on a spread of small problems (z to 1e-8, iteration counts)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import mgb_amd as m
from oracle import mgb_oracle as O

cases = []
for L in (2, 3, 4, 5):
    for p in (1.0, 1.3, 2.0, 3.5):
        cases.append((f"fem2d_P2 L={L} p={p}", lambda L=L, p=p: m.assemble(m.amg(m.subdivide(m.fem2d_P2(), L)), p=p)))
for L in (3, 5, 7):
    for p in (1.0, 2.5):
        cases.append((f"fem1d L={L} p={p}", lambda L=L, p=p: m.assemble(m.amg(m.subdivide(m.fem1d(), L)), p=p)))
for L in (1, 2, 3):
    cases.append((f"fem3d L={L} p=2", lambda L=L: m.assemble(m.amg(m.subdivide(m.fem3d(k=1), L)), p=2.0)))
for L in (2, 3):
    cases.append((f"fem2d geometric L={L}", lambda L=L: m.assemble(m.geometric_mg(m.fem2d_P2(), L), p=1.5)))
bad = 0
for name, make in cases:
    prob = make()
    t = time.time()
    try:
        sol = m.mgb_solve(prob)
        ref = O.mgb_solve(prob)
        err = float(np.abs(sol.z - ref["z"]).max())
        its_d, its_o = int(np.sum(sol.SOL_main["its"])), int(np.sum(ref["SOL_main"]["its"]))
        ok = err < 1e-8
        bad += (not ok)
        print(f"{name:28s} err {err:.2e} its device/oracle {its_d}/{its_o} {'ok' if ok else 'MISMATCH'} {time.time()-t:.1f}s", flush=True)
    except Exception as e:
        bad += 1
        print(f"{name:28s} EXCEPTION {type(e).__name__}: {str(e)[:100]}", flush=True)
print("failures:", bad)
