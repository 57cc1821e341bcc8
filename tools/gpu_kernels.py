"""Microbenchmark of the evaluate/assemble kernels at the fine level (hipEvent stage timers)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mgb_amd as m
from mgb_amd.device import DeviceMGBProblem
L = int(sys.argv[1]); p = float(sys.argv[2]); reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
kw = json.loads(sys.argv[4]) if len(sys.argv) > 4 else {"max_coarse": 300}
prob = m.assemble(m.amg(m.subdivide(m.fem2d_P2(), L), prolongator=m.amg_ruge_stuben(**kw)), p=p)
D = DeviceMGBProblem(prob); P = D.main
n = prob.M[0].w.size; J = len(P.level_sizes) - 1
z0 = np.ascontiguousarray(prob.g.T).reshape(-1); c = 0.1 * prob.f; s = np.zeros(P.level_sizes[J])
P.f0(J, s, c, z0); P.f1(J, s, c, z0); P.f2(J, s, c, z0, want_matrix=False)
P.reset_stage_timers(True)
for _ in range(reps):
    P.f0(J, s, c, z0); P.f1(J, s, c, z0); P.f2(J, s, c, z0, want_matrix=False)
B = dict(f0=219, f1=231 - 11.4, restrict=11.4 * 2 + 24, f2=347, assemble=488)   # SURVEY section 8(d) bytes / node
# the Newton loop's own f2: from the second call on the element kernel condenses the leaves (kernels.hpp)
if os.environ.get("MGB_NEWTON_F2", "1") == "1":
    for _ in range(reps + 1):
        P.newton_direction(J, s, c, z0)
for st in ('f0', 'f1', 'restrict', 'f2', 'assemble'):
    ms, cnt = P.stage_ms(st)
    us = 1e3 * ms / max(cnt, 1)
    gbs = B[st] * n / (us * 1e-6) / 1e9
    print(f"{st:9s} {us:8.1f} us  {gbs:8.1f} GB/s  {100*gbs/8000:5.1f}% of 8 TB/s", flush=True)
D.close()
