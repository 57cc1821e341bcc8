import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mgb_amd as m
from mgb_amd.device import DeviceMGBProblem
from mgb_amd.solve import mgb_driver
from mgb_amd.amg_prolongators import amg_ruge_stuben
L = int(sys.argv[1]); p = float(sys.argv[2]); kw = json.loads(sys.argv[3]) if len(sys.argv) > 3 else {}
factory = m.amg_smoothed_aggregation if os.environ.get('SA') else amg_ruge_stuben
t0 = time.time()
prob = m.assemble(m.amg(m.subdivide(m.fem2d_P2(), L), prolongator=factory(**kw)), p=p)
print('L', L, 'p', p, kw, 'setup %.1f' % (time.time() - t0), 'levels', [R.shape[1] for R in prob.M[0].R_fine], flush=True)
D = DeviceMGBProblem(prob)
t0 = time.time()
try:
    SOL = mgb_driver(D)
    if os.environ.get('TWICE'):
        print('  first solve %.2fs (includes plan + symbolic analysis)' % SOL['SOL_main']['t_elapsed'], flush=True)
        SOL = mgb_driver(D)
    sm = SOL['SOL_main']
    print('  OK its', int(sm['its'].sum()), sm['its'].sum(axis=1).tolist(), 'tsteps', sm['k'], 'core %.2fs' % sm['t_elapsed'], 'solve_s %.2f' % sm['solve_seconds'], flush=True)
except Exception as e:
    print('  FAILED', str(e)[:100], '%.1fs' % (time.time() - t0), flush=True)
D.close()
