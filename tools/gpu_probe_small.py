"""Probe build only (make CXXFLAGS+=-DMGB_STEP_PROBE): phase timestamps of one level-1 front in mf_factor_small."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mgb_amd as m
from mgb_amd.device import DeviceMGBProblem
prob = m.assemble(m.amg(m.subdivide(m.fem2d_P2(), 9)), p=1.0)
D = DeviceMGBProblem(prob); P = D.main
J = len(P.level_sizes) - 1
z0 = np.ascontiguousarray(prob.g.T).reshape(-1); c = 0.1 * prob.f; s = np.zeros(P.level_sizes[J])
g = P.f1(J, s, c, z0); P.f2(J, s, c, z0, want_matrix=False)
P.solve(J, g)
out = (C.c_longlong * 64)()
P.lib.mgbhip_debug_probe(out)
P.solve(J, g)
P.lib.mgbhip_debug_probe(out)
v = np.array(out[:])
print("mf_factor_small level-1 front, us since entry: zero+A, extend-add, panels, write-back:", [round((v[40 + i] - v[40]) * 0.01, 2) for i in range(1, 5)])
D.close()
