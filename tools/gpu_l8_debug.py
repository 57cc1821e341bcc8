"""Debug helper: the reference-default hierarchy at L=8, p=1.5 (expected: 'Initial centering failed')."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mgb_amd as m
from mgb_amd.solve import MGBConvergenceFailure
prob = m.assemble(m.amg(m.subdivide(m.fem2d_P2(), 8)), p=1.5)
t = time.time()
try:
    m.mgb_solve(prob)
    print("converged?!", time.time() - t)
except MGBConvergenceFailure as e:
    print("failure:", e.code, str(e)[:200], time.time() - t)
