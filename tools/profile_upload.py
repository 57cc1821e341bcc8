"""cProfile of DeviceMGBProblem(prob) at L = 9 (time-to-first-solution work: where the 'upload' seconds go)."""
import cProfile, pstats, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mgb_amd as m
from mgb_amd.device import DeviceMGBProblem
prob = m.assemble(m.amg(m.subdivide(m.fem2d_P2(), 9)), p=1.0)
pr = cProfile.Profile(); t = time.time(); pr.enable()
D = DeviceMGBProblem(prob, device_id=0)
pr.disable(); print("upload", time.time() - t)
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
D.close()
