"""Phase timestamps of the backward sweep on the root front (probe build: -DMGB_STEP_PROBE, MGBHIP_LIB=...)."""
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
P.solve(J, g); P.solve(J, g)
out = (C.c_longlong * 64)()
P.lib.mgbhip_debug_probe(out)
v = np.array(out[:])
names = ["entry", "t gathered", "boundary rows done", "loop start", "step j0=256 start", "product done", "after sync 1", "update done", "step j0=224 start", "end"]
for i, nm in enumerate(names):
    print(f"{nm:22s} {(v[i] - v[0]) * 0.01:8.2f} us")
D.close()
