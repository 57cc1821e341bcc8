"""fem3d() Q1 p=4 (BASELINE configs[3] family) on one GPU: python tools/gpu_fem3d.py L [p]"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import mgb_amd as m
import json
L = int(sys.argv[1]); p = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
kw = json.loads(sys.argv[3]) if len(sys.argv) > 3 else {}
t0 = time.time()
prob = m.assemble(m.amg(m.subdivide(m.fem3d(k=1), L), prolongator=m.amg_ruge_stuben(**kw)), p=p)
print(f"fem3d L={L} p={p}: setup {time.time()-t0:.1f}s, levels {[R.shape[1] for R in prob.M[0].R_fine]}", flush=True)
for rep in range(int(os.environ.get('REPS', '2'))):
    t0 = time.time()
    sol = m.mgb_solve(prob)
    its = int(sol.SOL_main["its"].sum()) + (int(sol.SOL_feasibility["its"].sum()) if sol.SOL_feasibility else 0)
    tt = sol.SOL_main["t_elapsed"] + (sol.SOL_feasibility["t_elapsed"] if sol.SOL_feasibility else 0.0)
    print(f"rep {rep}: phaseI={'yes' if sol.SOL_feasibility else 'no'} its {its} wall {time.time()-t0:.2f}s core {tt:.2f}s it/s {its/tt:.1f}", flush=True)
