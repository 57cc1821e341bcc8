"""A/B wall-clock of warm solves without stage timers: python tools/gpu_ab.py L p reps [rs kwargs json]
(MGBHIP_LIB selects the build).  Prints iteration count, min / median wall per solve."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mgb_amd as m
from mgb_amd.device import DeviceMGBProblem
from mgb_amd.solve import mgb_driver
L = int(sys.argv[1]); p = float(sys.argv[2]); reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
kw = json.loads(sys.argv[4]) if len(sys.argv) > 4 else {}
prob = m.assemble(m.amg(m.subdivide(m.fem2d_P2(), L), prolongator=m.amg_ruge_stuben(**kw)), p=p)
D = DeviceMGBProblem(prob)
S = mgb_driver(D)
z1 = S["z"].copy()
ts = []
for _ in range(reps):
    t0 = time.perf_counter(); S = mgb_driver(D); ts.append(time.perf_counter() - t0)
its = int(S["SOL_main"]["its"].sum())
ts.sort()
print(f"{os.environ.get('MGBHIP_LIB', 'default lib')}: its {its} wall min {ts[0]:.4f} median {ts[len(ts)//2]:.4f} s -> {its/ts[0]:.1f} it/s (best), bitwise repeat {np.array_equal(z1, S['z'])}")
D.close()
