"""Factor/solve microbenchmark at the fine level: python tools/gpu_solver_bench.py L p reps
."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mgb_amd as m
from mgb_amd.device import DeviceMGBProblem
L = int(sys.argv[1]); p = float(sys.argv[2]); reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
prob = m.assemble(m.amg(m.subdivide(m.fem2d_P2(), L), prolongator=m.amg_ruge_stuben(max_coarse=300)), p=p)
D = DeviceMGBProblem(prob); P = D.main
J = len(P.level_sizes) - 1
z0 = np.ascontiguousarray(prob.g.T).reshape(-1); c = 0.1 * prob.f; s = np.zeros(P.level_sizes[J])
g = P.f1(J, s, c, z0); P.f2(J, s, c, z0, want_matrix=False)
try:
    P.solve(J, g)
except Exception as e:
    print("solve status:", str(e)[:60])
P.reset_stage_timers(True)
for _ in range(reps):
    try:
        P.solve(J, g)
    except Exception:
        pass
for st in ('factor', 'trisolve'):
    ms, cnt = P.stage_ms(st)
    print(f"{st:9s} {1e3*ms/max(cnt,1):9.1f} us", flush=True)
D.close()
