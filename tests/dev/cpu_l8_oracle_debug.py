"""Debug helper (CPU, imports the oracle: development only): the oracle on the reference-default hierarchy at L=8, p=1.5."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import mgb_amd as m
from oracle import mgb_oracle as O
prob = m.assemble(m.amg(m.subdivide(m.fem2d_P2(), 8)), p=1.5)
st = {"deadline": time.perf_counter() + float(sys.argv[1]) if len(sys.argv) > 1 else 250.0}
orig_dot = np.dot
t = time.time()
try:
    O.mgb_solve(prob, stats=st)
    print("converged?!", time.time() - t)
except O.MGBConvergenceFailure as e:
    print("failure:", e.code, str(e)[:160], "its", st.get("newton_its"), "s", time.time() - t)
except TimeoutError:
    print("timeout: its", st.get("newton_its"), "s", time.time() - t)
