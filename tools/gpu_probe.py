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
probe0 = (C.c_longlong * 64)()
P.lib.mgbhip_debug_probe(probe0)   # resets the min/max slots
P.solve(J, g)

out = (C.c_longlong * 64)()
P.lib.mgbhip_debug_probe(out)
v = np.array(out[:])
print("look-ahead WG (us since entry):", [round((v[i] - v[0]) * 0.01, 2) for i in range(8)])
print("tile 0 WG     (us since entry):", [round((v[16 + i] - v[16]) * 0.01, 2) for i in range(5)])
print("tile0 entry - LA entry (us):", (v[16] - v[0]) * 0.01)
print("LDLT shader cycles:", v[33] - v[32], "-> clock GHz ~", (v[33] - v[32]) / ((v[5] - v[4]) * 10.0))
D.close()
print("LDLT clock at columns 0,8,16,24 (cycles since col 0):", [int(v[40+i]-v[40]) for i in range(4)], "end:", int(v[33]-v[40]), "start->col0:", int(v[40]-v[32]))
print("fwd_inv level-6 front (us since entry): after init gather, W ready, wave0 done, rows done, end:", [round((v[48+i]-v[48])*0.01,2) for i in range(1,6)])
print("fwd_inv level-6 kernel: first start -> last end (us):", (v[57]-v[56])*0.01)
print("shader clock over the look-ahead workgroup: %d cycles in %.2f us -> %.2f GHz" % (v[35] - v[34], (v[7] - v[0]) * 0.01, (v[35] - v[34]) / ((v[7] - v[0]) * 10.0)))
print("32x32 LDLt in the look-ahead workgroup, first run: %d cycles / %.2f us; second run (same code, same data): %d cycles / %.2f us" % (v[33] - v[32], (v[37] - v[36]) * 0.01, v[39] - v[38], (v[45] - v[44]) * 0.01))
print("absolute (us since LA entry): LDLt#1 start %.2f end %.2f, LDLt#2 start %.2f end %.2f" % tuple((v[i] - v[0]) * 0.01 for i in (36, 37, 44, 45)))
