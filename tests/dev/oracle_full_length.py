"""Full-length oracle runs on the reference-default ladder (development only, imports the oracle).

    python tests/dev/oracle_full_length.py fem2d_P2 8 1.5 [max_coarse]   > tests/dev/logs/<name>.log

Runs the oracle's `mgb_solve` with the reference's own limits (maxit = 10000 in the initial centring's
J-j == 1 Newton solves, src/mgb.jl:64-73) and logs every Newton solve: level size, iterations, outcome,
and every 50th iteration (objective, lambda^2, cond(H) for systems of <= 64 unknowns)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import mgb_amd as m
from oracle import mgb_oracle as O

if os.environ.get("ORACLE_SOLVER") == "mf":      # host multifrontal Cholesky instead of SuperLU (3-D meshes: SuperLU's fill is prohibitive)
    O.set_solver("mf")
fam, L, p = sys.argv[1], int(sys.argv[2]), float(sys.argv[3])
kw = {}
if len(sys.argv) > 4:
    kw["prolongator"] = m.amg_ruge_stuben(max_coarse=int(sys.argv[4]))
geo = {"fem2d_P2": lambda: m.fem2d_P2(), "fem3d": lambda: m.fem3d(k=1), "fem1d": lambda: m.fem1d()}[fam]()
prob = m.assemble(m.amg(m.subdivide(geo, L), **kw), p=p)
print("ladder", [R.shape[1] for R in prob.M[0].R_fine], flush=True)
t0 = time.time()
state = {"solve": 0, "last_m": None, "k": 0}


def trace(k, y, inc, H, g, n):
    msz = g.size
    if k == 1:
        state["solve"] += 1
        print(f"[{time.time()-t0:8.1f}s] newton #{state['solve']} m={msz}", flush=True)
    if k % 50 == 0 or inc <= 0 or k <= 3:
        cond = ""
        if msz <= 64:
            Hd = H.toarray() if hasattr(H, "toarray") else np.asarray(H)
            ev = np.linalg.eigvalsh(0.5 * (Hd + Hd.T))
            cond = f" eig[{ev[0]:.3e},{ev[-1]:.3e}]"
        print(f"   k={k:5d} y={y:.15e} lambda2={inc:.6e} |g|={np.linalg.norm(g):.3e}{cond}", flush=True)
    state["k"] = k


st = {"trace": trace}
try:
    sol = O.mgb_solve(prob, stats=st)
    its = sol["SOL_main"]["its"]
    print("CONVERGED its per level", its.sum(axis=1).tolist(), "total", int(its.sum()),
          "feas", None if sol["SOL_feasibility"] is None else int(sol["SOL_feasibility"]["its"].sum()),
          f"{time.time()-t0:.1f}s", flush=True)
except O.MGBConvergenceFailure as e:
    print("FAILURE", e.code, str(e)[:200], "newton_its", st.get("newton_its"), f"{time.time()-t0:.1f}s", flush=True)
