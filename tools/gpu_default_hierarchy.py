"""Default-hierarchy probe: run mgb_solve with amg_ruge_stuben(max_coarse=2) (the reference default,
src/multigrid.jl:296) at the BASELINE sizes and report levels, iteration counts or the failure."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mgb_amd as m
from mgb_amd.device import DeviceMGBProblem
from mgb_amd.solve import mgb_driver

out = []
for (L, p) in [(7, 1.5), (8, 1.0), (8, 1.5), (9, 1.0), (9, 1.5)]:
    t0 = time.time()
    prob = m.assemble(m.amg(m.subdivide(m.fem2d_P2(), L)), p=p)
    rec = dict(L=L, p=p, setup_s=round(time.time() - t0, 2), levels=[int(R.shape[1]) for R in prob.M[0].R_fine])
    D = DeviceMGBProblem(prob)
    t0 = time.time()
    try:
        SOL = mgb_driver(D)
        sm = SOL['SOL_main']
        rec.update(ok=True, its=int(sm['its'].sum()), its_per_level=sm['its'].sum(axis=1).tolist(), tsteps=int(sm['k']),
                   core_s=round(float(sm['t_elapsed']), 3), first_step_its=sm['its'][:, 0].tolist())
    except Exception as e:
        rec.update(ok=False, error=str(e)[:200], seconds=round(time.time() - t0, 2))
    D.close()
    print(json.dumps(rec), flush=True)
    out.append(rec)
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out', 'default_hierarchy.json'), 'w'), indent=1)
