"""Per-tree-level factor/solve timing at the fine level (MGBHIP_LEVEL_TIMING=1):
python tools/gpu_solver_levels.py L p reps [rs kwargs json]"""
import sys, os, time, json
os.environ["MGBHIP_LEVEL_TIMING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mgb_amd as m
from mgb_amd.device import DeviceMGBProblem
L = int(sys.argv[1]); p = float(sys.argv[2]); reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
kw = json.loads(sys.argv[4]) if len(sys.argv) > 4 else {"max_coarse": 300}
prob = m.assemble(m.amg(m.subdivide(m.fem2d_P2(), L), prolongator=m.amg_ruge_stuben(**kw)), p=p)
D = DeviceMGBProblem(prob); P = D.main
J = len(P.level_sizes) - 1
z0 = np.ascontiguousarray(prob.g.T).reshape(-1); c = 0.1 * prob.f; s = np.zeros(P.level_sizes[J])
g = P.f1(J, s, c, z0); P.f2(J, s, c, z0, want_matrix=False)
x0 = P.solve(J, g)
P.reset_stage_timers(True)
for _ in range(reps):
    x = P.solve(J, g)
assert np.array_equal(x, x0)
H = P.f2(J, s, c, z0)
print("residual", float(np.linalg.norm(H @ x - g) / np.linalg.norm(g)))
tot = {}
for st in ('factor', 'trisolve'):
    ms, cnt = P.stage_ms(st)
    tot[st] = 1e3 * ms / max(cnt, 1)
    print(f"{st:9s} {tot[st]:9.1f} us", flush=True)
for pre in ('fac', 'fwd', 'bwd'):
    row = []
    for lv in range(40):
        ms, cnt = P.stage_ms(f"{pre}_lv{lv:02d}")
        if cnt == 0:
            break
        row.append(1e3 * ms / cnt)
    print(pre, " ".join(f"{v:7.1f}" for v in row), " sum %.1f" % sum(row), flush=True)
print(json.dumps(P.solver_stats(J)))
D.close()
