"""Config 4 (fem3d Q1, p = 4, L = 6, max_coarse=500) with the stage timers on: iteration counts per level of both phases, stage
averages, and per-tree-level factor / backward times (MGBHIP_LEVEL_TIMING=1).  Usage: python tools/gpu_fem3d_stages.py [L] [p] [max_coarse]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mgb_amd as m
from mgb_amd.solve import mgb_driver

L = int(sys.argv[1]) if len(sys.argv) > 1 else 6
p = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
mc = int(sys.argv[3]) if len(sys.argv) > 3 else 500
prob = m.assemble(m.amg(m.subdivide(m.fem3d(k=1), L), prolongator=m.amg_ruge_stuben(max_coarse=mc)), p=p)
sol = m.mgb_solve(prob, keep_device=True)
D = sol.device
for img in (D.main, D.feasibility):
    img.reset_stage_timers(True)
t0 = time.perf_counter(); S = mgb_driver(D); dt = time.perf_counter() - t0
print("levels", D.main.level_sizes, "wall %.3f s" % dt)
for key, img in (("SOL_feasibility", D.feasibility), ("SOL_main", D.main)):
    if S[key] is None:
        continue
    its = S[key]["its"]
    print(key, "its", int(its.sum()), "per level", its.sum(axis=1).tolist() if its.ndim == 2 else its.tolist(), "core %.3f s" % S[key]["t_elapsed"])
    for st in ("f0", "f1", "f01", "f2", "assemble", "f0_coarse", "f1_coarse", "f01_coarse", "f2_coarse", "assemble_coarse", "restrict", "prolong", "factor", "trisolve"):
        ms, cnt = img.stage_ms(st)
        if cnt:
            print(f"  {st:16s} n={cnt:5d} avg {1e3*ms/cnt:8.1f} us  total {ms:8.1f} ms")
    if os.environ.get("MGBHIP_LEVEL_TIMING") == "1":
        for pre in ("fac", "bwd"):
            row = []
            for lv in range(40):
                ms, cnt = img.stage_ms(f"{pre}_lv{lv:02d}")
                if cnt == 0:
                    break
                row.append(f"{ms:.1f}/{cnt}")
            print(" ", pre, "total ms / launches per tree level:", " ".join(row))
D.close()
