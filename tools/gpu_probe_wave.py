"""Phase timestamps of one wave of the level-1 factor kernel (probe build: -DMGB_STEP_PROBE, MGBHIP_LIB=...)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mgb_amd as m
from mgb_amd.device import DeviceMGBProblem
prob = m.assemble(m.amg(m.subdivide(m.fem2d_P2(), 9)), p=1.0)
D = DeviceMGBProblem(prob); P = D.main
J = len(P.level_sizes) - 1
z0 = np.ascontiguousarray(prob.g.T).reshape(-1); c = 0.1 * prob.f; s = np.zeros(P.level_sizes[J])
for _ in range(3):
    P.newton_direction(J, s, c, z0)
out = (C.c_longlong * 64)()
P.lib.mgbhip_debug_probe(out)
v = np.array(out[:])
names = ["entry", "zeroed, first loads back", "A scattered", "children added", "rows in registers", "LDLt done", "stored"]
for i, nm in enumerate(names):
    print(f"{nm:26s} {(v[8 + i] - v[8]) * 0.01:8.2f} us")
for base, lvl in ((40, "level 2 (4096 fronts, m=40 k=7)"), (24, "level 4 (1024 fronts, m=80 k=15)")):
    print("LDS-front kernel,", lvl)
    for i, nm in enumerate(["entry", "A scattered", "children added", "factored", "stored"]):
        print(f"   {nm:22s} {(v[base + i] - v[base]) * 0.01:8.2f} us")
D.close()
print("level-4 front, first block step: ", [round((v[48 + i] - v[48]) * 0.01, 2) for i in range(7)],
      "= start, wave-0 LDLt done, barrier, row solves done, barrier, update done, barrier")
