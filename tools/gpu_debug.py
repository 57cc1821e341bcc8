import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mgb_amd as m
from mgb_amd.device import DeviceMGBProblem
from mgb_amd.solve import mgb_driver
L = int(sys.argv[1]); p = float(sys.argv[2])
t0 = time.time()
prob = m.assemble(m.amg(m.subdivide(m.fem2d_P2(), L)), p=p)
print('setup', time.time() - t0, flush=True)
D = DeviceMGBProblem(prob)
t0 = time.time()
try:
    SOL = mgb_driver(D)
    sm = SOL['SOL_main']
    print('OK its', int(sm['its'].sum()), sm['its'].sum(axis=1), 'core', sm['t_elapsed'], 'solve_s', sm['solve_seconds'], flush=True)
except Exception as e:
    print('FAILED', e, time.time() - t0, flush=True)
for J in range(len(D.main.level_sizes)):
    try:
        print(J, D.main.solver_stats(J))
    except Exception as e:
        print(J, 'nostats', e)
