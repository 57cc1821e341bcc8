"""First-light GPU check: per-primitive parity against the oracle, then golden solves."""
import json, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, scipy.sparse as sp
import mgb_amd as m
from mgb_amd.device import DeviceMGBProblem
from oracle import mgb_oracle as O

def rel(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))

def prim_check(name, prob, tval=0.1):
    D = DeviceMGBProblem(prob)
    M = O.OracleAMG(prob.M[0]) if not isinstance(prob.M[0].D_fine[0], np.ndarray) else O.OracleAMG(prob.M[0])
    B = O.Barrier(prob.Q)
    z0 = np.ascontiguousarray(prob.g.T).reshape(-1).copy()
    c = tval * prob.f
    rng = np.random.default_rng(1)
    worst = 0
    for J in range(len(M.R_fine)):
        R = M.R_fine[J]
        s = 1e-3 * rng.standard_normal(R.shape[1])
        y_o = B.f0(s, M.w, c, R, M.D_fine, z0)
        g_o = B.f1(s, M.w, c, R, M.D_fine, z0)
        H_o = sp.csr_matrix(B.f2(s, M.w, c, R, M.D_fine, z0))
        y_d = D.main.f0(J, s, c, z0)
        g_d = D.main.f1(J, s, c, z0)
        H_d = D.main.f2(J, s, c, z0)
        x_d = D.main.solve(J, g_d)
        x_o = O.solve_symmetric(sp.csc_matrix(H_o), g_o)
        e = [abs(y_d - y_o) / abs(y_o), rel(g_d, g_o), float(abs(H_d - H_o).max() / abs(H_o).max()), rel(x_d, x_o)]
        worst = max(worst, max(e))
        print(f"  {name} level {J} m={R.shape[1]} f0 {e[0]:.2e} f1 {e[1]:.2e} f2 {e[2]:.2e} solve {e[3]:.2e}", flush=True)
    D.close()
    return worst

def build(c):
    g = c['geom']
    if g == 'fem1d': geom = m.fem1d(nodes=np.linspace(-1, 1, c['nodes']))
    elif g == 'fem2d_P2': geom = m.subdivide(m.fem2d_P2(), c['L'])
    elif g == 'fem3d': geom = m.subdivide(m.fem3d(k=c['k']), c['L'])
    elif g == 'spectral1d': geom = m.spectral1d(n=c['n'])
    elif g == 'spectral2d': geom = m.spectral2d(n=c['n'])
    else: return None
    return m.assemble(m.amg(geom), p=c['p'])

if __name__ == "__main__":
    out = {}
    print("== primitive parity ==", flush=True)
    out['prim_fem2d_P2_L3'] = prim_check('fem2d_P2 L3 p1.5', m.assemble(m.amg(m.subdivide(m.fem2d_P2(), 3)), p=1.5))
    out['prim_fem1d'] = prim_check('fem1d 9 nodes', m.assemble(m.amg(m.fem1d(nodes=np.linspace(-1, 1, 9))), p=2.0))
    out['prim_fem3d'] = prim_check('fem3d k1 L2', m.assemble(m.amg(m.subdivide(m.fem3d(k=1), 2)), p=1.5), tval=0.1)
    print("== golden solves ==", flush=True)
    d = json.load(open(os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden', 'golden.json')))
    for c in d['cases']:
        prob = build(c)
        if prob is None: continue
        t0 = time.time()
        try:
            sol = m.mgb_solve(prob)
            gold = np.array(c['z_colmajor']).reshape(2, -1).T
            err = float(np.linalg.norm(sol.z - gold))
            its = int(sol.SOL_main['its'].sum())
            print(f"  {c['name']}: err={err:.3e} its={its} phase1={'y' if sol.SOL_feasibility else 'n'} {time.time()-t0:.2f}s", flush=True)
            out[c['name']] = err
        except Exception as e:
            import traceback; traceback.print_exc()
            print(f"  {c['name']}: FAILED {e}", flush=True)
            out[c['name']] = str(e)
    os.makedirs('gpurun_out', exist_ok=True)
    json.dump(out, open('gpurun_out/first_light.json', 'w'), indent=1)
