"""Scale check on the GPU: end-to-end solves with stage timers; oracle comparison at small L."""
import json, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import mgb_amd as m
from mgb_amd.device import DeviceMGBProblem
from mgb_amd.solve import mgb_driver

def run(L, p, compare=False):
    t0 = time.time()
    prob = m.assemble(m.amg(m.subdivide(m.fem2d_P2(), L)), p=p)
    t_setup = time.time() - t0
    t0 = time.time()
    D = DeviceMGBProblem(prob)
    t_up = time.time() - t0
    D.main.reset_stage_timers(True)
    t0 = time.time()
    SOL = mgb_driver(D)
    t_solve = time.time() - t0
    sm = SOL['SOL_main']
    its = int(sm['its'].sum())
    stages = {s: D.main.stage_ms(s) for s in ('f0', 'f1', 'f2', 'assemble', 'factor', 'trisolve', 'restrict', 'prolong')}
    nl = len(D.main.level_sizes)
    st = D.main.solver_stats(nl - 1)
    print(f"L={L} p={p}: n={prob.M[0].w.size} m_fine={D.main.level_sizes[-1]} setup {t_setup:.1f}s upload {t_up:.2f}s "
          f"solve wall {t_solve:.2f}s core {sm['t_elapsed']:.3f}s its={its} -> {its/sm['t_elapsed']:.1f} it/s "
          f"solve_s={sm['solve_seconds']:.3f} f0={sm['f0_evals']} f1={sm['f1_evals']} f2={sm['f2_evals']}", flush=True)
    print("   its per level:", sm['its'].sum(axis=1).tolist(), " tsteps", sm['k'], flush=True)
    print("   stages(ms total, launches):", {k: (round(v[0], 2), v[1]) for k, v in stages.items()}, flush=True)
    print("   fine solver:", st, flush=True)
    rec = dict(L=L, p=p, its=its, core_s=sm['t_elapsed'], solve_s=sm['solve_seconds'], stages=stages, solver=st)
    if compare:
        from oracle import mgb_oracle as O
        t0 = time.time()
        so = O.mgb_solve(prob)
        print(f"   oracle: its={int(so['SOL_main']['its'].sum())} {time.time()-t0:.1f}s  max|z_gpu - z_oracle| = {np.abs(SOL['z']-so['z']).max():.3e}", flush=True)
        rec['zdiff'] = float(np.abs(SOL['z'] - so['z']).max())
    D.close()
    return rec

if __name__ == "__main__":
    cases = json.loads(sys.argv[1]) if len(sys.argv) > 1 else [[4, 1.5, 1], [5, 1.0, 1], [6, 1.5, 0], [7, 1.5, 0]]
    out = []
    for L, p, cmp_ in cases:
        try:
            out.append(run(L, p, bool(cmp_)))
        except Exception as e:
            import traceback; traceback.print_exc()
            print("FAILED", L, p, e, flush=True)
    os.makedirs('gpurun_out', exist_ok=True)
    json.dump(out, open('gpurun_out/scale.json', 'w'), indent=1, default=str)
