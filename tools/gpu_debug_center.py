import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mgb_amd as m
from mgb_amd.device import DeviceMGBProblem
from mgb_amd.solve import mgb_driver
L = int(sys.argv[1]); p = float(sys.argv[2])
prob = m.assemble(m.amg(m.subdivide(m.fem2d_P2(), L)), p=p)
print('levels', [int(R.shape[1]) for R in prob.M[0].R_fine], flush=True)
D = DeviceMGBProblem(prob)
try:
    SOL = mgb_driver(D)
    print('OK', int(SOL['SOL_main']['its'].sum()))
except Exception as e:
    print('FAILED', str(e)[:200])
D.close()
