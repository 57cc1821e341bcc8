"""Quick headline check: python tools/gpu_bench_quick.py L p [rs kwargs json] -> its, wall per solve (2nd solve), stage times"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mgb_amd as m
from mgb_amd.device import DeviceMGBProblem
from mgb_amd.solve import mgb_driver
L = int(sys.argv[1]); p = float(sys.argv[2]); kw = json.loads(sys.argv[3]) if len(sys.argv) > 3 else {}
prob = m.assemble(m.amg(m.subdivide(m.fem2d_P2(), L), prolongator=m.amg_ruge_stuben(**kw)), p=p)
D = DeviceMGBProblem(prob)
S = mgb_driver(D)
z1 = S["z"].copy()
D.main.reset_stage_timers(True)
t0 = time.perf_counter(); S = mgb_driver(D); dt = time.perf_counter() - t0
its = S["SOL_main"]["its"]
print("its", int(its.sum()), "per level", its.sum(axis=1).tolist(), "wall %.3f s" % dt, "it/s %.1f" % (its.sum() / dt), "bitwise repeat", np.array_equal(z1, S["z"]))
for st in ("f0", "f1", "f01", "f2", "assemble", "f0_coarse", "f1_coarse", "f01_coarse", "f2_coarse", "assemble_coarse", "restrict", "prolong", "factor", "trisolve"):
    ms, cnt = D.main.stage_ms(st)
    if cnt:
        print(f"  {st:16s} n={cnt:5d} avg {1e3*ms/cnt:8.1f} us  total {ms:8.1f} ms")
if os.environ.get("MGBHIP_LEVEL_TIMING") == "1":
    for pre in ("fac", "bwd"):
        row = []
        for lv in range(40):
            ms, cnt = D.main.stage_ms(f"{pre}_lv{lv:02d}")
            if cnt == 0:
                break
            row.append(f"{ms:.1f}/{cnt}")
        print(pre, "total ms / launches per tree level:", " ".join(row))
D.close()
