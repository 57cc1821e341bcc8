"""cProfile of amg() at fem2d_P2 level L sorted by cumulative time.  Usage: python tools/amg_profile.py [L] [top]"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import mgb_amd as m

L = int(sys.argv[1]) if len(sys.argv) > 1 else 9
top = int(sys.argv[2]) if len(sys.argv) > 2 else 45
g = m.subdivide(m.fem2d_P2(), L)
pr = cProfile.Profile()
t = time.perf_counter()
pr.enable()
mg = m.amg(g)
pr.disable()
print(f"== amg: {time.perf_counter() - t:.3f} s")
pstats.Stats(pr).sort_stats("cumulative").print_stats(top)
