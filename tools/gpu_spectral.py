"""Time the dense (spectral) path: python tools/gpu_spectral.py n p  (BASELINE configs[4] family)."""
import sys, time, json
import numpy as np
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import mgb_amd as m

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
p = float(sys.argv[2]) if len(sys.argv) > 2 else 1.5
t0 = time.time()
mg = m.amg(m.spectral2d(n=n))
nn = mg.geometry.w.size
if os.environ.get("OBSTACLE", "1") != "0":
    # BASELINE configs[4]: two-sided obstacle (reference pattern: src/Zoo/two_sided_obstacle.jl:23-49)
    Q = m.intersect(mg, m.convex_Euclidian_power(mg, idx=(2, 3, 4), p_grid=np.full(nn, p)),
                    m.convex_linear(mg, idx=(1,), A=lambda x: np.array([[1.0], [-1.0]]), b=lambda x: np.array([0.1, 1.0])))
    prob = m.assemble(mg, Q=Q, f_grid=np.tile([2.0, 0, 0, 0.5], (nn, 1)), g_grid=np.tile([0.0, 10.0], (nn, 1)))
else:
    prob = m.assemble(mg, p=p)
print("setup", round(time.time() - t0, 2), "s; levels", [R.shape for R in prob.M[0].R_fine], flush=True)
for rep in range(2):
    t0 = time.time()
    sol = m.mgb_solve(prob)
    dt = time.time() - t0
    its = int(sol.SOL_main["its"].sum())
    print(f"rep {rep}: its {its} wall {dt:.3f}s core {sol.SOL_main['t_elapsed']:.3f}s  it/s {its/sol.SOL_main['t_elapsed']:.1f}", flush=True)
st = sol.stage_ms if hasattr(sol, "stage_ms") else None
print(json.dumps(st) if st else "")
